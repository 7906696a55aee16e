"""The CPU oracle (oracle/*.py) against the committed golden vectors (tools/make_golden.py:
transformers.MPNetModel / BertModel outputs + the transformers bucket / position-id functions)."""
import numpy as np
import pytest

from arxiv_rag_amd import config as C
from arxiv_rag_amd.weights import seeded_state_dict
from oracle import encoder_oracle as EO
from oracle import search_oracle as SO


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize("name", ["tiny-mpnet", "tiny-bert", "tiny-bert-cls"])
def test_tiny_all_intermediates(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    cfg = C.PRESETS[name]
    sd = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    ids, lens = g["ids"], g["lens"]
    h, allh = EO.encoder_hidden(sd, cfg, ids, lens, return_all=True)
    valid = (np.arange(ids.shape[1])[None] < lens[:, None])
    for i, x in enumerate(allh):
        ref = g[f"hidden_{i}"]
        # padded query rows are defined-but-unused in both; compare valid tokens only
        assert np.abs(x - ref)[valid].max() < 2e-5, (name, i)
    pooled = EO.pool(h, lens, cfg.pool)
    assert np.abs(pooled - g["pooled"]).max() < 2e-5
    emb = EO.l2_normalize(pooled)
    assert np.abs(emb - g["emb"]).max() < 2e-6
    assert _cos(emb, g["emb"]).min() > 1 - 1e-6


def test_tiny_weights_regenerate_from_seed(golden_dir):
    """seeded_state_dict is the bit-stable generator the full-shape fixtures rely on."""
    from tools.make_golden import WSPEC
    g = np.load(golden_dir / "tiny-mpnet.npz")
    sd = seeded_state_dict(C.TINY_MPNET, seed=11, **WSPEC)
    for k, v in sd.items():
        assert np.array_equal(v, g["w:" + k]), k


def test_batch_composition_independence(golden_dir):
    """with a key-padding mask a text's embedding does not depend on its batch."""
    g = np.load(golden_dir / "tiny-mpnet.npz")
    cfg = C.TINY_MPNET
    sd = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    ids, lens = g["ids"], g["lens"]
    for r in (1, 4, 5):
        solo = EO.encode_tokens(sd, cfg, ids[r:r + 1, :lens[r]], lens[r:r + 1])
        assert _cos(solo[0], g["emb"][r]) > 1 - 1e-6


@pytest.mark.parametrize("key", ["all-MiniLM-L6-v2:w05", "all-MiniLM-L6-v2:hf02"])
def test_full_minilm_shape(golden_dir, key):
    _check_full(golden_dir, key, "all-MiniLM-L6-v2")


@pytest.mark.parametrize("key", ["all-mpnet-base-v2:w05"])
def test_full_mpnet_shape(golden_dir, key):
    _check_full(golden_dir, key, "all-mpnet-base-v2", rows=[0, 2, 5, 7, 14])


def _check_full(golden_dir, key, name, rows=None):
    g = np.load(golden_dir / "full_shapes.npz")
    cfg = C.PRESETS[name]
    seed, std, bstd, jit = g[key + ":wspec"]
    sd = seeded_state_dict(cfg, seed=int(seed), std=std, bias_std=bstd, ln_jitter=jit)
    ids, lens, ref = g[key + ":ids"], g[key + ":lens"], g[key + ":emb"]
    rows = list(range(len(lens))) if rows is None else rows
    seqs = [ids[r, :lens[r]].tolist() for r in rows]
    emb = EO.encode_ragged(sd, cfg, seqs, batch_size=8)
    c = _cos(emb, ref[rows])
    assert c.min() > 1 - 1e-6, c
    assert np.abs(emb - ref[rows]).max() < 5e-6


@pytest.mark.parametrize("name,rows", [("all-MiniLM-L6-v2", None), ("all-mpnet-base-v2", [1, 6, 9, 14])])
def test_full_shapes_adversarial_statistics(golden_dir, name, rows):
    """The oracle on `weights.adversarial_state_dict` (wide LayerNorm affine with outlier channels, rows whose mean dwarfs their
    spread, heavy-tailed matrices) against the transformers outputs in full_shapes_adv.npz, and the weights' digest."""
    import hashlib
    from arxiv_rag_amd.weights import adversarial_state_dict
    g = np.load(golden_dir / "full_shapes_adv.npz")
    cfg = C.PRESETS[name]
    key = name + ":adv"
    seed, off = g[key + ":wspec"]
    sd = adversarial_state_dict(cfg, seed=int(seed), row_offset=float(off))
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode()); h.update(np.ascontiguousarray(sd[k]).tobytes())
    assert h.digest() == bytes(g[key + ":wdigest"])                      # the weights regenerate bit-exactly from the seed
    ids, lens, ref = g[key + ":ids"], g[key + ":lens"], g[key + ":emb"]
    rows = list(range(len(lens))) if rows is None else rows
    emb = EO.encode_ragged(sd, cfg, [ids[r, :lens[r]].tolist() for r in rows], batch_size=8)
    assert _cos(emb, ref[rows]).min() > 1 - 1e-5
    assert np.abs(emb - ref[rows]).max() < 2e-4


def test_relative_position_buckets(golden_dir):
    g = np.load(golden_dir / "mpnet_tables.npz")
    assert np.array_equal(EO.relative_position_bucket(g["delta"]), g["bucket_of_delta"])
    for S in (8, 256, 384, 512):
        d0 = np.arange(S) - 0
        assert np.array_equal(EO.relative_position_bucket(d0), g[f"bucket_S{S}_row0"])
        dl = np.arange(S) - (S - 1)
        assert np.array_equal(EO.relative_position_bucket(dl), g[f"bucket_S{S}_rowlast"])
    # spot values recorded in SURVEY.md §8c(iii)
    r0 = g["bucket_S256_row0"]
    assert r0[0] == 0 and r0[1] == 17 and r0[-1] == 31 and g["bucket_S256_rowlast"][0] == 15


def test_toeplitz_table_equals_dense_bias():
    cfg = C.TINY_MPNET
    sd = seeded_state_dict(cfg, seed=3)
    for S in (5, 64, 300):
        dense = EO.position_bias(sd, cfg, S)
        t = EO.toeplitz_bias_table(sd, cfg, S)
        i, j = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
        assert np.array_equal(dense, t[:, j - i + S - 1])


def test_position_ids(golden_dir):
    g = np.load(golden_dir / "mpnet_tables.npz")
    assert np.array_equal(EO.mpnet_position_ids(g["posid_ids"], 1), g["posid_expected"])


def test_search_golden(golden_dir):
    g = np.load(golden_dir / "search_4096x768.npz")
    n, d, sc, nq, sq = g["recipe"]
    Cm = SO.unit_rows_f16(int(n), int(d), int(sc)); Q = SO.unit_rows_f16(int(nq), int(d), int(sq))
    Cm[100] = Cm[17]; Cm[2000] = Cm[17]; Cm[3000] = Cm[17]; Q[0] = Cm[17]
    s, i = SO.topk_search(Cm, Q, 10, block=1000)          # odd block: exercises the blocked merge
    assert np.array_equal(i, g["ids"])
    assert np.allclose(s, g["scores"], atol=1e-6)
    assert i[0, :4].tolist() == [17, 100, 2000, 3000]      # exact ties -> lower index first


def test_search_sharded_merge_equals_global():
    Cm = SO.unit_rows_f16(3000, 64, 1); Q = SO.unit_rows_f16(9, 64, 2)
    s, i = SO.topk_search(Cm, Q, 10)
    parts = [SO.topk_search(Cm[a:b], Q, 10, idx_base=a) for a, b in ((0, 700), (700, 1500), (1500, 3000))]
    ms, mi = SO.merge_partials(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 10)
    assert np.array_equal(mi, i) and np.array_equal(ms, s)


def test_search_k_larger_than_n():
    Cm = SO.unit_rows_f16(4, 32, 1); Q = SO.unit_rows_f16(2, 32, 2)
    s, i = SO.topk_search(Cm, Q, 10)
    assert (i[:, 4:] == -1).all() and np.isneginf(s[:, 4:]).all() and (i[:, :4] >= 0).all()


def test_oracle_vs_transformers_direct():
    """The numpy restatement against the third-party modules themselves (oracle/tf_reference.py), fresh inputs."""
    pytest.importorskip("transformers")
    from oracle import tf_reference as TF
    for cfg in (C.TINY_MPNET, C.TINY_BERT_CLS):
        sd = seeded_state_dict(cfg, seed=21, std=0.06, bias_std=0.03, ln_jitter=0.1)
        rs = np.random.RandomState(2)
        lens = np.array([40, 7, 1, 64, 33], np.int64)
        ids = np.full((5, 64), cfg.pad_id, np.int64)
        for r, n in enumerate(lens):
            ids[r, :n] = rs.randint(4, cfg.vocab_size, size=n)
        ref = TF.encode_tokens(TF.build_model(cfg, sd), cfg, ids, lens)
        got = EO.encode_tokens(sd, cfg, ids, lens)
        assert np.abs(got - ref).max() < 5e-6
