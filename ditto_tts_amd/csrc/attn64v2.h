// attn64v2.h — the attn64v2 forward-attention kernel (head_dim 64), shared by two translation units: attention.hip (the
// inference instantiations) and attention_train.hip (the TRAIN / DROP instantiations, compiled with -fno-slp-vectorize: their
// few extra products would otherwise be paired into v_pk_mul_f32, and packed fp32 instructions occupy the matrix pipe).
// Include inside `namespace ditto { namespace {` after attn_common.h.
#pragma once

constexpr int DH = ATT_DH, QBLK = 128, KBLK = ATT_KBLK;
constexpr int KV_TILE_BYTES = ATT_KV_TILE_BYTES;  // 8 KiB
constexpr float RESCALE_THR_LOG2 = ATT_RESCALE_THR_LOG2;   // attn_common.h

// ------------------------------------------------------------------------------------------------
// attn64v2: attn64 with the softmax's VALU work cut by ~40 %.  The kernel is VALU-ISSUE-bound at d_h = 64 (per wave
// and 64-key tile ~240 issue slots of 4 cycles against 16 MFMAs; two waves share a SIMD's issue port), so every
// instruction removed from the tile loop is time.  Contract: q arrives PRE-SCALED by scale*log2(e) (folded into the
// q rows of the packed in-projection weights at ditto_model_create), so S is already in log2 units, and:
//   * S' = S - m comes straight out of the MFMA chain: the first MFMA of a chain takes a constant accumulator block
//     holding -m (rewritten only when the running maximum moves) => no per-element scale/subtract (-32 v_fma);
//   * the row sum l = sum_k P is an MFMA with an all-ones A operand on the packed P (4 MFMAs per tile into one
//     accumulator block whose rows are all equal) => no per-element adds (-32 v_add) and no lane^32 exchange for l;
//   * K/V DMA source addresses are base + tile * stride (the per-tile clamp only exists on a ragged last tile).
// Deferred raise of the maximum as in attn64: P <= 2^8.
// ------------------------------------------------------------------------------------------------
// NBUF = 2: K/V one tile ahead, waited with vmcnt(0) at the end of a tile (grids of many workgroups per CU: the other
// workgroups cover the DMA latency).  NBUF = 4 (64 KiB): K/V THREE tiles ahead behind a counted vmcnt, for small grids
// (batch-1 serving: 96 workgroups on 256 CUs, where the kernel's time is one workgroup's serial tile loop and a single
// tile of look-ahead exposes the full DMA latency on every tile: 1.3 us per 64-key tile measured).
// TRAIN (the training forward): q arrives UNSCALED and its fragments are scaled here, once per kernel, to bf16(q * scale *
// log2(e)) — the attention backward's dq kernel scales its query fragments the same way, so its recomputed scores are these
// bit for bit; the log2-domain log-sum-exp m + log2(l) of every query row goes to p.lse.  DROP: train-mode dropout on the
// probabilities (hash mask, common.h): the O += V P product takes the masked, rescaled P, the row sum l stays that of the
// full softmax.  TRAIN / DROP instantiations live in attention_train.hip (no SLP packing of their extra products; an inline-asm
// multiply is not an option: the compiler's hazard recognizer does not see an asm statement read a transcendental's result).
#ifdef DITTO_DIAG_A2_STAMP   // tools/build_diag_one.sh ... attention.hip -DDITTO_DIAG_A2_STAMP: where does a wave's 64-key tile spend its cycles?
                             // s_memtime stamps around the five segments of tile_body, summed per wave in scalar registers (timing build
                             // only: every stamp drains the LDS reads in flight and pins the instruction order — read the SHARES).
constexpr int A2_STAMP_WAVES = 3072 * 4;
__device__ unsigned long long g_a2_stamps[A2_STAMP_WAVES * 8];
DITTO_DEV unsigned long long a2_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define A2_STAMP(i) do { const unsigned long long _t = a2_now(); a2_acc[i] += _t - a2_prev; a2_prev = _t; } while (0)
#else
#define A2_STAMP(i)
#endif

template <bool RESID, int WPS = 2, int NBUF = 2, bool TRAIN = false, bool DROP = false>
__global__ __launch_bounds__(256, WPS) void attn64v2_kernel(AttnParams p) {
    constexpr bool PFV = WPS <= 2;   // V fragments prefetched ahead of the softmax only when 256 registers are available
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * KV_TILE_BYTES];  // [buf][K|V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    int qrow = qb * QBLK + wid * 32 + ql;
    const bool qvalid = qrow < p.Sq;
    qrow = qvalid ? qrow : p.Sq - 1;

    bf16x8 qf[4];
    {
        const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
        if constexpr (TRAIN) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) qf[ks][i] = (bf16)((float)qf[ks][i] * p.scale_log2);
        }
    }
    DropStream dstream{};
    if constexpr (DROP) dstream = drop_stream(p.seed_lo, p.seed_hi, p.layer, bh);
    const int nkt = (p.Skv + KBLK - 1) / KBLK;
    const bool ragged = (p.Skv & (KBLK - 1)) != 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // this lane's two (row, chunk) DMA sources of tile 0; tile kt is + kt * 64 rows
    const bf16 *ksrc[2], *vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3), cpos = lane & 7;
        ksrc[i] = p.k + ((size_t)b * p.Skv + row) * p.ldk + h * DH + (cpos ^ ((row >> 1) & 7)) * 8;
        vsrc[i] = p.v + ((size_t)b * p.Skv + row) * p.ldv + h * DH + (cpos ^ (((row >> 1) & 1) << 2)) * 8;
    }
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    auto dma_kv = [&](int kt, int buf) {
        if (ragged && kt == nkt - 1) {   // rows past Skv are clamped (never read out of bounds), masked in the tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wid * 2 + i;
                const int row = piece * 8 + (lane >> 3), cpos = lane & 7;
                int key = kt * KBLK + row;
                key = key < p.Skv ? key : p.Skv - 1;
                const int ck = cpos ^ ((row >> 1) & 7), cv = cpos ^ (((row >> 1) & 1) << 2);
                glds16(p.k + ((size_t)b * p.Skv + key) * p.ldk + h * DH + ck * 8,
                       lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + piece * 1024));
                glds16(p.v + ((size_t)b * p.Skv + key) * p.ldv + h * DH + cv * 8,
                       lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            glds16(ksrc[i] + (size_t)kt * kstep, lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + piece * 1024));
            glds16(vsrc[i] + (size_t)kt * vstep,
                   lds_base + (unsigned)(buf * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
        }
    };

    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 ot[2], lsum, cneg;   // cneg: every register = -m_run (the MFMA chains' initial accumulator)
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; lsum[i] = 0.f; cneg[i] = 0.f; }

    if constexpr (NBUF == 2) {
        dma_kv(0, 0);
#ifdef DITTO_DIAG_A2_NODMA   // both buffers hold tile 0: every later tile computes on VALID data (no NaN garbage that would speed the
        dma_kv(0, 1);        // whole step up through the clock), only the traffic is gone
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
#pragma unroll
        for (int t0 = 0; t0 < NBUF - 1; ++t0)
            if (t0 < nkt) dma_kv(t0, t0);
    }

#ifdef DITTO_DIAG_A2_STAMP
    unsigned long long a2_acc[6] = {0, 0, 0, 0, 0, 0}, a2_prev = a2_now();
#endif
    auto tile_body = [&](int kt, auto MASKED) {
        A2_STAMP(5);   // loop back (and, on the first tile, the prologue)
        const char* kb = smem + (kt % NBUF) * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;
        if constexpr (NBUF == 2) {
#ifndef DITTO_DIAG_A2_NODMA   // tools/build_diag.sh: attn64v2 without its K/V tile traffic after tile 0 (TIMING ONLY)
            if (kt + 1 < nkt) dma_kv(kt + 1, (kt + 1) & 1);
#endif
        } else {
            // tile kt has landed once at most the loads of the tiles behind it (4 per wave each) are in flight
            const int ahead = nkt - 1 - kt;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // every wave's pieces of tile kt are visible; every wave is done with tile kt-1's buffer
            if (kt + NBUF - 1 < nkt) dma_kv(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
        }

        // ---- S'^T[key][query] = K Q'^T - m  (log2 units) ----
        f32x16 st[2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + kb2 * 32 * 128 + k_row_off +
                                                                   (((2 * ks + hh) ^ k_swz) << 4));
                st[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? cneg : st[kb2], 0, 0, 0);
            }
        A2_STAMP(0);   // next tile's DMA issued, K fragments read, the 8 S MFMAs issued
        if constexpr (decltype(MASKED)::value) {
            const int kbase_idx = kt * KBLK + 4 * hh;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_idx + kb2 * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= p.Skv) st[kb2][r] = -1e30f;
                }
        }
        bf16x8 vf[8];
        if constexpr (PFV) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                    const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                    vf[s2 * 2 + db] = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                           __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                }
            __builtin_amdgcn_sched_barrier(0);
        }

#if defined(DITTO_DIAG_A2_NOSOFTMAX)   // TIMING ONLY: no running maximum, no exponentials — MFMAs, LDS reads, DMA and the bf16 packing alone
        bf16x8 pf[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = (bf16)st[s2 >> 1][8 * (s2 & 1) + j];
#else
        // ---- row maximum relative to the running one; raise it (rarely) ----
        float dm = fmaxf(st[0][0], st[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) dm = fmaxf(dm, fmaxf(st[0][r], st[1][r]));
        dm = fmaxf(dm, __shfl_xor(dm, 32, 64));
        if (kt == 0 || !__all(dm <= RESCALE_THR_LOG2)) {
            // first tile: the running maximum IS this tile's.  Whole octaves: the rescale factors are powers of two, which is
            // what lets attn64v3 rescale a bf16 P exactly and stay BITWISE equal to this kernel
            const float up = ceilf(kt == 0 ? dm : fmaxf(dm, 0.f));
            const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                st[0][i] -= up; st[1][i] -= up; cneg[i] -= up;
                ot[0][i] *= alpha; ot[1][i] *= alpha; lsum[i] *= alpha;
            }
        }
        A2_STAMP(1);   // S back from the matrix pipe, row maximum, lane exchange, the (rare) raise
        // ---- P = exp2(S');  O^T += V^T P^T ;  l += 1^T P^T ----
        // DROP: group by group (16 keys): the masked, rescaled probabilities pd feed O += V P and die at once, the row sum l
        // takes the full softmax's p — and the next group's exponentials and hashes issue behind this group's MFMAs
        bf16x8 pf[DROP ? 1 : 4];
        if constexpr (!DROP) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#ifdef DITTO_DIAG_A2_EXPADD   // TIMING ONLY: the transcendental replaced by one plain full-rate operation (the SFU's share of the tile)
                    pf[s2][j] = (bf16)(st[s2 >> 1][8 * (s2 & 1) + j] + 1.0f);
#else
                    pf[s2][j] = (bf16)__builtin_amdgcn_exp2f(st[s2 >> 1][8 * (s2 & 1) + j]);
#endif
                }
        }
#endif   // DITTO_DIAG_A2_NOSOFTMAX
        A2_STAMP(2);   // 32 exponentials + packing to bf16
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            bf16x8 pl, pv;
            if constexpr (DROP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rr = 8 * (s2 & 1) + j;
                    const int key = kt * KBLK + (s2 >> 1) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
                    const float e = __builtin_amdgcn_exp2f(st[s2 >> 1][rr]);
                    const float km = drop_keep(dstream, qrow, key, p.drop_thr) ? p.keep_scale : 0.f;
                    pl[j] = (bf16)e;
                    pv[j] = (bf16)(e * km);
                }
            } else {
                pl = pf[s2];
                pv = pf[s2];
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                if constexpr (PFV) {
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2 * 2 + db], pv, ot[db], 0, 0, 0);
                } else {
                    const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                    const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                    const bf16x8 vfr = cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                    ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pv, ot[db], 0, 0, 0);
                }
            }
            lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pl, lsum, 0, 0, 0);
        }
        A2_STAMP(3);   // V fragments read, 8 + 4 MFMAs issued
        if constexpr (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        A2_STAMP(4);   // the next tile's DMA landed, the workgroup's barrier
    };
    const int nfull = ragged ? nkt - 1 : nkt;
    for (int kt = 0; kt < nfull; ++kt) tile_body(kt, std::false_type{});
    if (ragged) tile_body(nkt - 1, std::true_type{});

#ifdef DITTO_DIAG_A2_STAMP
    {
        const int w = blockIdx.x * 4 + wid;
        if (lane == 0 && w < A2_STAMP_WAVES) {
            for (int i = 0; i < 6; ++i) g_a2_stamps[w * 8 + i] = a2_acc[i];
            g_a2_stamps[w * 8 + 6] = (unsigned long long)nkt;
            g_a2_stamps[w * 8 + 7] = 1;
        }
    }
#endif
    const float inv = 1.0f / lsum[0];
    if (!qvalid) return;
    if constexpr (TRAIN) {   // log2-domain log-sum-exp of the scaled scores: running maximum (= -cneg) + log2(row sum)
        if (p.lse && hh == 0) p.lse[(size_t)bh * p.Sq + qrow] = __builtin_amdgcn_logf(lsum[0]) - cneg[0];
    }
    const size_t grow = (size_t)b * p.Sq + qrow;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = h * DH + 32 * db + 8 * g + 4 * hh;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[db][4 * g + e] * inv;
            if constexpr (RESID) {
                attn_resid_update(p, grow, col, o);
                if constexpr (TRAIN) {   // training on the bf16 stream: O itself for the backward (h_after - h_before of two bf16 rows is not O)
                    if (p.out) {
                        u32x2 st2;
                        st2[0] = pack_bf16x2(o[0], o[1]);
                        st2[1] = pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
                    }
                }
            } else {
                u32x2 st2;
                st2[0] = pack_bf16x2(o[0], o[1]);
                st2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
            }
        }
}
