#!/usr/bin/env python3
"""CPU sanitizer fuzz of the native WordPiece feeder (GPU sanitizers are not available on the pool; this part of the C ABI is host code).
build:  g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared arxiv_rag_amd/csrc/wordpiece.cpp -o /tmp/libwp_asan.so
run:    LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python tools/wp_asan_fuzz.py
Random vocabulary, random byte soup (ASCII control bytes, multi-byte UTF-8, long runs), several max_len and thread counts, the
miss -> cache_add loop with random pieces; checks the output invariants (specials, padding, lengths, flags)."""
import ctypes as C, numpy as np, random
lib = C.CDLL("/tmp/libwp_asan.so")
lib.arx_wp_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int32]*1 + [C.c_int32]*6 + [C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
lib.arx_wp_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
lib.arx_wp_miss_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
lib.arx_wp_miss_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.arx_wp_cache_add.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
lib.arx_wp_destroy.argtypes = [C.c_void_p]
def blob(bs):
    off = np.zeros(len(bs)+1, np.int64); off[1:] = np.cumsum([len(b) for b in bs]) if bs else 0
    return b"".join(bs), off
rnd = random.Random(0)
letters = "abcdefghij"
vocab = [b"[PAD]", b"[UNK]", b"[CLS]", b"[SEP]"] + [c.encode() for c in letters] + [("##"+c).encode() for c in letters] + [b".", b",", b"!"]
for _ in range(300):
    w = "".join(rnd.choice(letters) for _ in range(rnd.randint(2, 9)))
    vocab.append((w if rnd.random() < 0.6 else "##"+w).encode())
vb, vo = blob(vocab)
tb, to = blob([b"[SEP]", b"<mask>"])
h = C.c_void_p()
assert lib.arx_wp_create(vb, vo.ctypes.data, len(vocab), 1, 2, 3, 0, 1, 100, tb, to.ctypes.data, 2, C.byref(h)) == 0
alphabet = [chr(c) for c in range(0, 128)] + ["é", "中", "́", "–", "\U0001d465", " "]
for it in range(200):
    n = rnd.randint(0, 400)
    texts = []
    for _ in range(n):
        L = rnd.choice([0, 1, 5, 50, 300, 2000])
        texts.append("".join(rnd.choice(alphabet if rnd.random() < 0.3 else letters + "   ") for _ in range(rnd.randint(0, L))).encode("utf-8"))
    bl, off = blob(texts)
    for max_len in (2, 3, 17, 384):
        ids = np.full((n, max_len), -7, np.int32); lens = np.full(n, -1, np.int32); fb = np.full(n, 9, np.uint8)
        assert lib.arx_wp_encode(h, bl, off.ctypes.data, n, max_len, ids.ctypes.data, lens.ctypes.data, fb.ctypes.data, rnd.choice([1, 3, 8])) == 0
        ok = fb == 0
        assert ((lens[ok] >= 2) & (lens[ok] <= max_len)).all() and (lens[~ok] == 0).all() and set(fb.tolist()) <= {0, 1, 2}
        for i in np.flatnonzero(ok)[:20]:
            assert ids[i, 0] == 2 and ids[i, lens[i]-1] == 3 and (ids[i, lens[i]:] == 0).all() and (ids[i, :lens[i]] >= 0).all()
        ns, nb = C.c_int64(), C.c_int64(); lib.arx_wp_miss_count(h, C.byref(ns), C.byref(nb))
        if ns.value:
            buf = C.create_string_buffer(max(1, nb.value)); o2 = np.zeros(ns.value+1, np.int64)
            lib.arx_wp_miss_fetch(h, buf, o2.ctypes.data)
            segs = [buf.raw[o2[i]:o2[i+1]] for i in range(ns.value)]
            pieces = [[rnd.randint(4, len(vocab)-1) for _ in range(rnd.randint(0, 6))] for _ in segs]
            sb, so = blob(segs); io = np.zeros(len(pieces)+1, np.int64); io[1:] = np.cumsum([len(p) for p in pieces])
            flat = np.array([t for p in pieces for t in p] or [0], np.int32)
            assert lib.arx_wp_cache_add(h, sb, so.ctypes.data, len(segs), flat.ctypes.data, io.ctypes.data) == 0
lib.arx_wp_destroy(h)
print("asan/ubsan fuzz ok")
