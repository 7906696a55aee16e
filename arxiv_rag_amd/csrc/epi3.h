// LDS-staged epilogue for the 256-wide GEMM kernels (gemm8.h).
#pragma once
#include "gemm.h"

// =====================================================================================================================
// Epilogue v3 (for the 256-wide kernels above): every per-column / per-row VECTOR the epilogue needs (bias, s_n, the
// residual LayerNorm's gamma/beta; row mean / rstd of the A operand and of the residual) is staged in LDS by ONE
// global_load_lds per wave at the very start of the tile (8 x 1 KB behind the k-tile buffers), so the epilogue issues no
// small global loads at all; the residual vectors ride a rolling window eight (column pair, row block) steps ahead of their use (one exposed
// latency instead of two, and no store drain in between: a vmcnt wait for the second pair's loads used to wait for the first
// pair's stores as well).  Same arithmetic as epilogue_store_v2.
struct EpiStage {
    enum { BIAS = 0, SVEC = 1, RGAMMA = 2, RBETA = 3, AMEAN = 4, ARSTD = 5, RMEAN = 6, RRSTD = 7, BYTES = 8 * 1024 };
};

template <int MODE>
__device__ __forceinline__ void epi_stage_issue(const EpiParams& p, int m0, int n0, char* stage, int wid, int lane, int N) {
    constexpr bool LN_IN = (MODE == EPI_LN_BIAS || MODE == EPI_LN_BIAS_GELU);
    constexpr bool LNR = (MODE == EPI_LNRESID_STATS);
    const int w = wid;                                           // wave-uniform (readfirstlane'd by the caller): scalar select and branch
    const float* src = nullptr;
    if (w == EpiStage::BIAS) src = p.bias + n0;
    if constexpr (LN_IN) {
        if (w == EpiStage::SVEC) src = p.s_vec + n0;
        if (w == EpiStage::AMEAN) src = p.a_sum + m0;
        if (w == EpiStage::ARSTD) src = p.a_sq + m0;
    }
    if constexpr (LNR) {
        if (w == EpiStage::RGAMMA) src = p.r_gamma + n0;
        if (w == EpiStage::RBETA) src = p.r_beta + n0;
        if (w == EpiStage::RMEAN) src = p.r_sum + m0;
        if (w == EpiStage::RRSTD) src = p.r_sq + m0;
    }
    // lane id recomputed by v_mbcnt (two VALU ops) rather than kept live or spilled across the k-loop
    unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    (void)lane;
    // a last n-tile that is only half there (N % 256 == 128): the column vectors end at N; lanes past it re-read the valid half
    if (w < EpiStage::AMEAN && n0 + 256 > N) l &= 31;
    if (src) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + l * 4), (lds_void_t*)(stage + w * 1024), 16, 0, 0);
}

// Full-line, non-temporal output stores (profiles/r03/gemm_epilogue_store_probe.md).  The 4 lanes of an output row hold 64 B of it per column
// pair: half a 128-B line per store instruction, the other half a step later.  Lanes mq and mq ^ 8 exchange one 16-B piece each (eight
// bank-masked DPP moves per row block), after which an instruction covers 8 rows x 128 B.  With complete lines the stores can be `nt`:
// the output stream (1.2-1.6 GB per launch through 32 MB of L2) then stops evicting operand lines (+50 % fabric reads with plain stores)
// and leaves fewer dirty lines for the next kernel to drain; `nt` on HALF lines is slower than plain stores (the halves leave the L2
// separately: +44 % fabric writes).  Same-box in-situ A/B, chunks/s: half-line plain 21 060 (placement only) · full-line plain except GELU
// 21 580 / 21 730 · + nt on those 21 870 · full-line nt everywhere 21 965.  ARX_FL_NONE / ARX_NT_NONE rebuild the older forms.
template <int MODE> struct EpiFullLine {
#if defined(ARX_FL_NONE)
    static constexpr bool value = false;
#else
    static constexpr bool value = true;
#endif
};
template <int MODE> struct EpiStoreNT {
#if defined(ARX_NT_NONE)
    static constexpr bool value = false;
#else
    static constexpr bool value = EpiFullLine<MODE>::value;
#endif
};

template <int MODE, bool CHECK, int DEPTH>
__device__ __forceinline__ void epilogue_store_v3_impl(const f32x4 (&acc)[4][8], const EpiParams& p, int m0, int n0, int wr, int wc,
                                                       int lane, int M, const char* stage) {
    constexpr int NI = 4, MI = 8, NS = (NI / 2) * MI;                 // NS (column pair, row block) steps; residual loads DEPTH steps ahead
    constexpr bool LN_IN = (MODE == EPI_LN_BIAS || MODE == EPI_LN_BIAS_GELU);
    constexpr bool STATS = (MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS);
    constexpr bool RESID = (MODE == EPI_BIAS_RESID || STATS);
    constexpr bool LNR = (MODE == EPI_LNRESID_STATS);
    constexpr bool FL = EpiFullLine<MODE>::value;
    const int mq = lane & 15, q = lane >> 4;
    const int m_base = m0 + wr * 128, n_base = n0 + wc * 64;
    const float* sf = reinterpret_cast<const float*>(stage);
    // row offsets are rebuilt per step from one base (block-uniform strides): the epilogue must stay well under 256 VGPRs or
    // the persistent kernel's loop-carried state gets spilled INTO the k-loop
    auto row_of = [&](int i) { int m = m_base + i * 16 + mq; if (CHECK) m = m < M ? m : M - 1; return m; };
    auto row_ok = [&](int i) { return !CHECK || (m_base + i * 16 + mq) < M; };
    // column inside the block tile: with the permuted W placement (gemm8.h b_slice_col) this lane's values of MFMA blocks 2jp and 2jp + 1
    // are the eight consecutive columns jp * 32 + q * 8 + [0, 8) — no cross-lane exchange, and the 4 lanes of a row cover 64 contiguous bytes
    auto col_of = [&](int jp) { return wc * 64 + jp * 32 + q * 8; };
    u32x4 rres[DEPTH];
    auto res_load = [&](int st) {
        const int i = st / (NI / 2), jp = st % (NI / 2);
        if (row_ok(i)) rres[st % DEPTH] = *reinterpret_cast<const u32x4*>(p.resid + (uint32_t)row_of(i) * (uint32_t)p.ldr + n0 + col_of(jp));
    };
    if constexpr (RESID) {
#pragma unroll
        for (int st = 0; st < DEPTH; ++st) res_load(st);
    }
    // row-block outer, column-pair inner: a row block's statistics close after two steps (2 live registers, not 16), and the
    // column vectors are simply re-read from the LDS stage each step (three to six ds_read_b128)
    float st_s = 0.f, st_q = 0.f;
    f32x2 bb[4], ss[4], gg[4], ee[4];
    u32x4 opair[2];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int i = st / (NI / 2), jp = st % (NI / 2);
        const int cb = col_of(jp);
        {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sf + EpiStage::BIAS * 256 + cb), b1 = *reinterpret_cast<const f32x4*>(sf + EpiStage::BIAS * 256 + cb + 4);
            bb[0] = f32x2{b0[0], b0[1]}; bb[1] = f32x2{b0[2], b0[3]}; bb[2] = f32x2{b1[0], b1[1]}; bb[3] = f32x2{b1[2], b1[3]};
            if constexpr (LN_IN) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sf + EpiStage::SVEC * 256 + cb), s1 = *reinterpret_cast<const f32x4*>(sf + EpiStage::SVEC * 256 + cb + 4);
                ss[0] = f32x2{s0[0], s0[1]}; ss[1] = f32x2{s0[2], s0[3]}; ss[2] = f32x2{s1[0], s1[1]}; ss[3] = f32x2{s1[2], s1[3]};
            }
            if constexpr (LNR) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(sf + EpiStage::RGAMMA * 256 + cb), g1 = *reinterpret_cast<const f32x4*>(sf + EpiStage::RGAMMA * 256 + cb + 4);
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(sf + EpiStage::RBETA * 256 + cb), e1 = *reinterpret_cast<const f32x4*>(sf + EpiStage::RBETA * 256 + cb + 4);
                gg[0] = f32x2{g0[0], g0[1]}; gg[1] = f32x2{g0[2], g0[3]}; gg[2] = f32x2{g1[0], g1[1]}; gg[3] = f32x2{g1[2], g1[3]};
                ee[0] = f32x2{e0[0], e0[1]}; ee[1] = f32x2{e0[2], e0[3]}; ee[2] = f32x2{e1[0], e1[1]}; ee[3] = f32x2{e1[2], e1[3]};
            }
        }
        const f32x4 lo = acc[2 * jp][i], hi = acc[2 * jp + 1][i];
        if (FL || row_ok(i)) {                                       // full-line stores: every lane computes, its piece may be stored by its partner lane
            const int rb = wr * 128 + i * 16 + mq;                            // row inside the block tile
            f32x2 vv[4] = {f32x2{lo[0], lo[1]}, f32x2{lo[2], lo[3]}, f32x2{hi[0], hi[1]}, f32x2{hi[2], hi[3]}};
            if constexpr (LN_IN) {
                const float am = sf[EpiStage::AMEAN * 256 + rb], ar = sf[EpiStage::ARSTD * 256 + rb];
                const f32x2 nm = f32x2{-am, -am}, rs = f32x2{ar, ar};
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = rs * (nm * ss[pi] + vv[pi]) + bb[pi];
            } else {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = vv[pi] + bb[pi];
            }
            if constexpr (MODE == EPI_BIAS_GELU || MODE == EPI_LN_BIAS_GELU) {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = gelu_poly_pk(vv[pi]);
            }
            if constexpr (RESID) {
                const u32x4 rr = rres[st % DEPTH];
                float rm = 0.f, rsd = 1.f;
                if constexpr (LNR) { rm = sf[EpiStage::RMEAN * 256 + rb]; rsd = sf[EpiStage::RRSTD * 256 + rb]; }
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) {
                    float r0, r1;
                    unpack_bf16x2(rr[pi], r0, r1);
                    f32x2 ra = f32x2{r0, r1};
                    if constexpr (LNR) ra = ((ra - f32x2{rm, rm}) * f32x2{rsd, rsd}) * gg[pi] + ee[pi];
                    vv[pi] = vv[pi] + ra;
                }
            }
            u32x4 o;
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) o[pi] = pack_bf16x2(vv[pi].x, vv[pi].y);
            if constexpr (FL) {
              opair[jp] = o;
              if (jp == NI / 2 - 1) {
                const bool up = mq >= 8;
                // bank-masked DPP moves write only the lanes that receive: lanes 8-15 of a row take their partner's jp = 1 piece into x,
                // lanes 0-7 their partner's jp = 0 piece into y (eight moves per row block, no selects)
                u32x4 x = opair[0], y = opair[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)x[r], (int)opair[1][r], 0x128, 0xf, 0xc, false);
                    y[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)y[r], (int)opair[0][r], 0x128, 0xf, 0x3, false);
                }
                const int cbs = col_of(up ? 1 : 0);
                const int rx = m_base + i * 16 + (mq & 7), ry = rx + 8;
                u32x4* dx = reinterpret_cast<u32x4*>(p.out + (uint32_t)rx * (uint32_t)p.ldc + n0 + cbs);
                u32x4* dy = reinterpret_cast<u32x4*>(p.out + (uint32_t)ry * (uint32_t)p.ldc + n0 + cbs);
#ifdef ARX_DEV_VARIANTS
                if (p.dev_store == 3) { asm volatile("" :: "v"(x), "v"(y), "v"(dx), "v"(dy)); if (p.ldc == 0x7fffffff) { *dx = x; *dy = y; } }
                else if (p.dev_store == 1) {      // asm: hipcc merged half of the builtin's nt stores with the plain branch's and dropped the flag
                    if (!CHECK || rx < M) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dx), "v"(x) : "memory");
                    if (!CHECK || ry < M) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dy), "v"(y) : "memory");
                }
                else
#endif
                if constexpr (EpiStoreNT<MODE>::value) {
                    if (!CHECK || rx < M) __builtin_nontemporal_store(x, dx);
                    if (!CHECK || ry < M) __builtin_nontemporal_store(y, dy);
                } else { if (!CHECK || rx < M) *dx = x; if (!CHECK || ry < M) *dy = y; }
              }
            } else {
                u32x4* dst = reinterpret_cast<u32x4*>(p.out + (uint32_t)row_of(i) * (uint32_t)p.ldc + n0 + cb);
#ifdef ARX_DEV_VARIANTS
                if (p.dev_store == 3) { asm volatile("" :: "v"(o), "v"(dst)); if (p.ldc == 0x7fffffff) *dst = o; }      // probe: epilogue math without its stores
                else if (p.dev_store == 1) __builtin_nontemporal_store(o, dst);
                else if (p.dev_store == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(o) : "memory");
                else
#endif
                if constexpr (EpiStoreNT<MODE>::value) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
            if constexpr (STATS) {
                // statistics of the bf16-ROUNDED values (what the consumers will read back)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x0, x1;
                    unpack_bf16x2(o[r], x0, x1);
                    st_s += x0 + x1;
                    st_q = fmaf(x0, x0, fmaf(x1, x1, st_q));
                }
            }
        }
        if constexpr (RESID) {
            if (st + DEPTH < NS) res_load(st + DEPTH);           // refill the slot just consumed
        }
        if constexpr (STATS) {
            if (jp == NI / 2 - 1) {                              // row block complete: this wave's 64-column partial
                float a = st_s, b = st_q;
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
                const int m = m_base + i * 16 + mq;
                const int64_t part = n_base >> 6;
                if (q == 0 && m < M) { p.o_sum[part * p.o_ld + m] = a; p.o_sq[part * p.o_ld + m] = b; }
                st_s = 0.f; st_q = 0.f;
            }
        }
    }
}

template <int MODE, int DEPTH = 8>
__device__ __forceinline__ void epilogue_store_v3(const f32x4 (&acc)[4][8], const EpiParams& p, int m0, int n0, int wr, int wc,
                                                  int lane, int M, int N, const char* stage) {
    if (n0 + wc * 64 >= N) return;            // N % 256 == 128: the right half of the last n-tile does not exist (wave-uniform)
    if (m0 + wr * 128 + 128 <= M) epilogue_store_v3_impl<MODE, false, DEPTH>(acc, p, m0, n0, wr, wc, lane, M, stage);
    else {
        epilogue_store_v3_impl<MODE, true, DEPTH>(acc, p, m0, n0, wr, wc, lane, M, stage);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // see epilogue_store_v2
    }
}
