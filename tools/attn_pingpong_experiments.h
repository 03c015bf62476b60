// attn_pingpong_experiments.h — round-6 EXPERIMENTS (not product code; included by tools/probe_attn64p.hip only): the attention forward
// as a barrier-separated ping-pong of a vector phase and a matrix phase, at two waves per SIMD (attn64pp, 64 queries per wave) and at
// four (attn32pp, 32 queries per wave).  Both are parity-correct and both measured SLOWER than attn64p (profiles/r06_attn_probe.txt):
// the vector work of a tile (64 exponentials, 64 adds, 32 converts, 32 maxima per 64 queries x 64 keys = ~1 200 issue cycles) is as
// long as its MFMAs (1 024), a wave alone in its vector phase issues one instruction per 4 cycles whatever the partner does, and a
// phase lasts ~2 000 cycles beside an MFMA-only partner.  Kept for the record of what was measured; include after attn64p.h.
#pragma once

// ------------------------------------------------------------------------------------------------
// attn64pp (round 6): attn64p's wave program — 64 queries per wave, every K / V fragment feeding two MFMAs — as a PING-PONG of the
// two waves of a SIMD.  Why: the knock-out builds of attn64p (tools/probe_attn64p.hip, profiles/r06_attn_probe.txt) price its
// softmax at a quarter of the launch although it is pure vector work beside an MFMA-only partner: two waves that run the SAME
// program fall into lockstep (guide, "two waves per SIMD" item 9) — both in their MFMA phase, then both in their vector phase —
// and the two pipes' times add (MFMA-only loop 2 560 cycles per pair of 64-query tiles, with the softmax 3 610).  Here the
// alternation is built in.  A workgroup is EIGHT waves (512 queries); waves 0..3 are group 0, waves 4..7 group 1 — the dispatcher
// puts wave w and wave w + 4 on the same SIMD — and a tile of a wave is two phases separated by workgroup barriers:
//     VV(t): row maxima, the (rare) raise, 64 exponentials, packing, row sums      — vector pipe only
//     MM(t): O += V(t) P(t) (16 MFMAs), then S'(t+1) = K(t+1) Q^T - m (16 MFMAs)    — matrix pipe + LDS fragment reads only
// with group 1 running ONE PHASE BEHIND group 0 (it waits at one extra barrier in front of its loop; group 0 at one behind its
// loop), so that between any two barriers a SIMD has one wave in VV and the other in MM:
//     barriers passed   0      1      2      3      4
//     group 0         VV(0)  MM(0)  VV(1)  MM(1)  VV(2) ...
//     group 1          --    VV(0)  MM(0)  VV(1)  MM(1) ...
// LDS: the query rows + K(0) + a ring of NB bundles {K(j+1), V(j)} = what MM(j) reads (8 + 16 NB KiB).  Bundle j is read in the phases 2j+1 (group 0)
// and 2j+2 (group 1); every wave moves one 1-KiB piece of each of its two tiles by LDS-DMA; bundle j + NB is issued into its slot
// at the start of phase 2j+3 (group 0: start of MM(j+1), group 1: start of VV(j+1)) and has until barrier 2(j+NB)+1 to land: each
// wave waits for its own pieces with a counted vmcnt in front of that barrier (group 0: the one that ends VV(j+NB); group 1: the one
// that ends MM(j+NB-1)), leaving the NB-2 younger bundles in flight.
// ------------------------------------------------------------------------------------------------
#ifndef DITTO_STATIC_FOR
#define DITTO_STATIC_FOR
template <typename F, int... I>
DITTO_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
DITTO_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#endif
// the last fragment (its index in the matrix phase's fetch order) that MFMA m of the phase reads
constexpr int mm_need(int m) { return m < 16 ? (m >> 1) : 8 + 4 * ((m - 16) >> 2) + (((m - 16) & 3) == 0 ? 1 : ((m - 16) & 3) == 1 ? 2 : 3); }

template <bool RESID, int NB = 4, int DIAG = 0>
__global__ __launch_bounds__(512, 2) void attn64pp_kernel(AttnParams p) {
    static_assert(NB >= 2 && NB <= 4, "ring depth");
    constexpr int QWG = 512;                                      // queries per workgroup (p.nqb counts blocks of this size)
    // [K(0)] [slot][K|V] [wave][its 64 query rows]: Q^T is read from LDS like K (same image, same swizzle) in the matrix phase, where
    // fragment reads are nearly free, instead of occupying 32 registers for the whole kernel
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE_BYTES * (1 + 2 * NB) + 8 * KV_TILE_BYTES];
    constexpr unsigned Q_OFF = KV_TILE_BYTES * (1 + 2 * NB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    int qrow[2];
    bool qvalid[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        qrow[x] = qb * QWG + wid * 64 + 32 * x + ql;
        qvalid[x] = qrow[x] < p.Sq;
        qrow[x] = qvalid[x] ? qrow[x] : p.Sq - 1;
    }
    const int nkt = p.Skv / KBLK;   // whole tiles (the launcher sends ragged key counts to attn64v2)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    {   // this wave's 64 query rows -> LDS, 8 pieces of 8 rows (rows past Sq clamped: their results are never stored)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 8 + (lane >> 3), cp = lane & 7;
            int qr = qb * QWG + wid * 64 + r;
            qr = qr < p.Sq ? qr : p.Sq - 1;
            glds16(p.q + ((size_t)b * p.Sq + qr) * p.ldq + h * DH + (cp ^ ((r >> 1) & 7)) * 8, lds_base + Q_OFF + wid * KV_TILE_BYTES + i * 1024);
        }
    }
    // this lane's (row, chunk) DMA source in tile 0 (LDS swizzles applied on the source); wave w moves piece w = rows 8w .. 8w+7
    const int drow = wid * 8 + (lane >> 3), dcp = lane & 7;
    const int dck = (dcp ^ ((drow >> 1) & 7)) * 8, dcv = (dcp ^ (((drow >> 1) & 1) << 2)) * 8;
    const bf16* ksrc = p.k + ((size_t)b * p.Skv + drow) * p.ldk + h * DH + dck;
    const bf16* vsrc = p.v + ((size_t)b * p.Skv + drow) * p.ldv + h * DH + dcv;
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    auto dma_k = [&](int tile, unsigned dst) { glds16(ksrc + (size_t)tile * kstep, lds_base + dst + wid * 1024); };   // this wave's piece
    auto dma_v = [&](int tile, unsigned dst) { glds16(vsrc + (size_t)tile * vstep, lds_base + dst + wid * 1024); };
    int issued = 0;   // bundles issued by this wave
    auto dma_bundle = [&](int j) {   // {K(j+1), V(j)} -> slot j % NB; always two loads (the last bundle re-reads K(nkt-1): unused)
        const unsigned slot = (unsigned)(KV_TILE_BYTES * (1 + 2 * (j % NB)));
        dma_k(j + 1 < nkt ? j + 1 : nkt - 1, slot);
        dma_v(j, slot + KV_TILE_BYTES);
        ++issued;
    };
    auto wait_bundle = [&](int j) {  // this wave's pieces of bundle j have landed: only the bundles issued after it may be in flight
        if constexpr (DIAG & 2) return;
        const int younger = issued - 1 - j;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto barrier = [&]() {
        if constexpr (!(DIAG & 4)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    };

    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 ot[2][2], cneg[2], st[2][2];
    float lrun[2] = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i) { ot[x][0][i] = 0.f; ot[x][1][i] = 0.f; cneg[x][i] = 0.f; }
    u32x4 pa[4], pb[4];

    // S'(tile) for both blocks from the K tile at LDS byte offset koff
    auto scores = [&](int tile, unsigned koff) {
        const char* kb = smem + koff;
        const char* qbase = smem + Q_OFF + wid * KV_TILE_BYTES;
        auto frag = [&](const char* base, int blk, int ks) {   // 32 rows x 16 head columns: row ql of block blk, 16-byte chunk 2 ks + hh
            return *reinterpret_cast<const bf16x8*>(base + blk * 32 * 128 + k_row_off + (((2 * ks + hh) ^ k_swz) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 q0 = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pa[ks]) : frag(qbase, 0, ks), q1 = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pb[ks]) : frag(qbase, 1, ks);
            const bf16x8 k0 = (DIAG & 8) ? q1 : frag(kb, 0, ks), k1 = (DIAG & 8) ? q0 : frag(kb, 1, ks);
            st[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, ks == 0 ? cneg[0] : st[0][0], 0, 0, 0);
            st[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1, ks == 0 ? cneg[1] : st[1][0], 0, 0, 0);
            st[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0, ks == 0 ? cneg[0] : st[0][1], 0, 0, 0);
            st[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, ks == 0 ? cneg[1] : st[1][1], 0, 0, 0);
        }
    };
    // VV(t): the vector phase
    auto softmax = [&](int t) {
        float dm[2] = {0.f, 0.f};
        if constexpr (!(DIAG & 1)) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float a = fmaxf(st[x][0][0], st[x][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) a = fmaxf(a, fmaxf(st[x][0][r], st[x][1][r]));
                float s0, s1;
                swap32(a, s0, s1);
                dm[x] = fmaxf(s0, s1);
            }
            if (t == 0 || !__all(fmaxf(dm[0], dm[1]) <= RESCALE_THR_LOG2)) {
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const float up = ceilf(t == 0 ? dm[x] : fmaxf(dm[x], 0.f));
#pragma unroll
                    for (int i = 0; i < 16; ++i) { st[x][0][i] -= up; st[x][1][i] -= up; cneg[x][i] -= up; }
                    if (t > 0) {
                        const float alpha = __builtin_amdgcn_exp2f(-up);
                        lrun[x] *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { ot[x][0][i] *= alpha; ot[x][1][i] *= alpha; }
                    }
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float e0 = st[x][s2 >> 1][8 * (s2 & 1) + 2 * j], e1 = st[x][s2 >> 1][8 * (s2 & 1) + 2 * j + 1];
                    if constexpr (!(DIAG & 1)) {
                        e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
                        if (j & 1) { l2 += e0; l3 += e1; } else { l0 += e0; l1 += e1; }
                    }
                    pk[j] = pack_bf16x2(e0, e1);
                }
                if (x == 0) pa[s2] = pk; else pb[s2] = pk;
            }
            lrun[x] += (l0 + l1) + (l2 + l3);
        }
    };
    // MM(t): the matrix phase — 32 MFMAs fed by 24 LDS fragment reads (8 V^T fragments, then per k-step K block 0, Q block A, Q block
    // B, K block 1), each read issued LEAD MFMAs ahead of its first use through a ring of registers and pinned there by scheduling
    // barriers: the partner wave is in its vector phase and issues no MFMA, so a fragment read that an MFMA waits for idles the pipe.
    // NEXT = false: the last tile (no scores to compute).  Every index is a compile-time constant (static_for).
    auto matrix = [&](int t, auto NEXT_) {
        constexpr bool NEXT = decltype(NEXT_)::value;
        const unsigned slot = (unsigned)(KV_TILE_BYTES * (1 + 2 * (t % NB)));
        const char* vb = smem + slot + KV_TILE_BYTES;
        const char* kb = smem + slot;
        const char* qbase = smem + Q_OFF + wid * KV_TILE_BYTES;
        constexpr int R = 12, LEAD = 5, NFR = NEXT ? 24 : 8, NM = NEXT ? 32 : 16;
        bf16x8 fr[R];
        auto fetch = [&](auto F_) {
            constexpr int f = decltype(F_)::value;
            if constexpr (f < 8) {
                constexpr int s2 = f >> 1, db = f & 1;
                const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                fr[f % R] = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pa[(s2 + db) & 3])
                                       : cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                              __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
            } else {
                constexpr int ks = (f - 8) >> 2, w = (f - 8) & 3;             // w: 0 = K block 0, 1 = Q block A, 2 = Q block B, 3 = K block 1
                const char* base = (w == 1 || w == 2 || (DIAG & 8)) ? qbase : kb;
                constexpr int blk = (w == 2 || w == 3) ? 1 : 0;
                fr[f % R] = *reinterpret_cast<const bf16x8*>(base + blk * 32 * 128 + k_row_off + (((2 * ks + hh) ^ k_swz) << 4));
            }
        };
        static_for<NM>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            constexpr int ahead = m + LEAD < NM - 1 ? m + LEAD : NM - 1;
            constexpr int lo = m == 0 ? 0 : mm_need(m - 1 + LEAD < NM - 1 ? m - 1 + LEAD : NM - 1) + 1, hi = mm_need(ahead) + 1;   // fetch [lo, hi)
            static_for<(hi > lo ? hi - lo : 0)>([&](auto D_) {
                if constexpr (lo + decltype(D_)::value < NFR) fetch(std::integral_constant<int, lo + decltype(D_)::value>{});
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < 16) {
                constexpr int f = m >> 1, s2 = f >> 1, db = f & 1, x = m & 1;
                ot[x][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[f % R], __builtin_bit_cast(bf16x8, x ? pb[s2] : pa[s2]), ot[x][db], 0, 0, 0);
            } else {
                constexpr int ks = (m - 16) >> 2, j = (m - 16) & 3, x = j & 1, kb2 = j >> 1;
                constexpr int fk = 8 + 4 * ks + (kb2 ? 3 : 0), fq = 8 + 4 * ks + 1 + x;
                st[x][kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[fk % R], fr[fq % R], ks == 0 ? cneg[x] : st[x][kb2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // the matrix phase as plain source order (the compiler's schedule): DIAG bit 5, the A/B partner of the hand-placed form
    auto matrix_simple = [&](int t) {
        const unsigned slot = (unsigned)(KV_TILE_BYTES * (1 + 2 * (t % NB)));
        const char* vb = smem + slot + KV_TILE_BYTES;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                const bf16x8 vf = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pa[(s2 + db) & 3])
                                             : cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                ot[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pa[s2]), ot[0][db], 0, 0, 0);
                ot[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pb[s2]), ot[1][db], 0, 0, 0);
            }
        if (t + 1 < nkt) scores(t + 1, slot);
    };

    // ---- prologue: K(0) and the first NB-1 bundles in flight; S'(0) ----
    dma_k(0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j < nkt) dma_bundle(j);
    {   // the query rows and K(0) are the oldest loads: everything issued after them may stay in flight
        if (issued >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (issued == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    barrier();
    scores(0, 0);
    if (grp == 1) {   // one phase behind: bundle 0 is first read in phase 1
        wait_bundle(0);
        barrier();
    }
    for (int t = 0; t < nkt; ++t) {
        const bool refill = t >= 1 && t - 1 + NB < nkt && !(DIAG & 2);
        if (grp == 1 && refill) dma_bundle(t - 1 + NB);
        if constexpr (DIAG & 64) __builtin_amdgcn_sched_barrier(0);
        softmax(t);
        if constexpr (DIAG & 64) __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) wait_bundle(t);
        barrier();
        if (grp == 0 && refill) dma_bundle(t - 1 + NB);
        if constexpr (DIAG & 64) __builtin_amdgcn_sched_barrier(0);
        if constexpr (DIAG & 32) matrix_simple(t);
        else if (t + 1 < nkt) matrix(t, std::true_type{});
        else matrix(t, std::false_type{});
        if constexpr (DIAG & 64) __builtin_amdgcn_sched_barrier(0);
        if (grp == 1 && t + 1 < nkt) wait_bundle(t + 1);
        barrier();
    }
    if (grp == 0) barrier();

    // ---- epilogue: normalise; lane (query ql, half hh) owns d = 32 db + 8 g + 4 hh + 0..3 of its two queries ----
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float s0, s1;
        swap32(lrun[x], s0, s1);
        const float l = s0 + s1;
        const float inv = 1.0f / l;
        if (!qvalid[x]) continue;
        const size_t grow = (size_t)b * p.Sq + qrow[x];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = h * DH + 32 * db + 8 * g + 4 * hh;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[x][db][4 * g + e] * inv;
                if constexpr (RESID) {
                    attn_resid_update(p, grow, col, o);   // x = attn_out + residual (src/components/DiT.py:139)
                } else {
                    u32x2 st2;
                    st2[0] = pack_bf16x2(o[0], o[1]);
                    st2[1] = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// attn32pp (round 6, experiment): the ping-pong at FOUR waves per SIMD — sixteen waves of 32 queries, two groups of eight; between
// two barriers every SIMD has two waves in the vector phase (the vector pipe retires an instruction every 2 cycles only when two
// waves feed it) and two in the matrix phase (whose fragment-read latencies cover one another).  128 registers per wave: one
// S' / P / O set, Q^T from LDS.  Waves 0..7 move the K pieces of a bundle, waves 8..15 its V pieces (one load per wave and bundle).
// ------------------------------------------------------------------------------------------------
template <bool RESID, int NB = 4, int DIAG = 0>
__global__ __launch_bounds__(1024, 4) void attn32pp_kernel(AttnParams p) {
    constexpr int QWG = 512;
    __shared__ __attribute__((aligned(16))) char smem[KV_TILE_BYTES * (1 + 2 * NB) + 8 * KV_TILE_BYTES];
    constexpr unsigned Q_OFF = KV_TILE_BYTES * (1 + 2 * NB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..15
    const int grp = wid >> 3;
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qb = id % p.nqb, bh = id / p.nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int ql = lane & 31, hh = lane >> 5;
    int qrow = qb * QWG + wid * 32 + ql;
    const bool qvalid = qrow < p.Sq;
    qrow = qvalid ? qrow : p.Sq - 1;
    const int nkt = p.Skv / KBLK;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // this wave's 32 query rows -> LDS
        const int r = i * 8 + (lane >> 3), cp = lane & 7;
        int qr = qb * QWG + wid * 32 + r;
        qr = qr < p.Sq ? qr : p.Sq - 1;
        glds16(p.q + ((size_t)b * p.Sq + qr) * p.ldq + h * DH + (cp ^ ((r >> 1) & 7)) * 8, lds_base + Q_OFF + wid * (KV_TILE_BYTES / 2) + i * 1024);
    }
    const int piece = wid & 7;
    const bool mover_k = wid < 8;
    const int drow = piece * 8 + (lane >> 3), dcp = lane & 7;
    const int dck = (dcp ^ ((drow >> 1) & 7)) * 8, dcv = (dcp ^ (((drow >> 1) & 1) << 2)) * 8;
    const bf16* ksrc = p.k + ((size_t)b * p.Skv + drow) * p.ldk + h * DH + dck;
    const bf16* vsrc = p.v + ((size_t)b * p.Skv + drow) * p.ldv + h * DH + dcv;
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    int issued = 0;
    auto dma_bundle = [&](int j) {   // {K(j+1), V(j)} -> slot j % NB: one load per wave
        const unsigned slot = (unsigned)(KV_TILE_BYTES * (1 + 2 * (j % NB)));
        if (mover_k) glds16(ksrc + (size_t)(j + 1 < nkt ? j + 1 : nkt - 1) * kstep, lds_base + slot + piece * 1024);
        else glds16(vsrc + (size_t)j * vstep, lds_base + slot + KV_TILE_BYTES + piece * 1024);
        ++issued;
    };
    auto wait_bundle = [&](int j) {
        if constexpr (DIAG & 2) return;
        const int younger = issued - 1 - j;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto barrier = [&]() {
        if constexpr (!(DIAG & 4)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    };
    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    f32x16 ot[2], cneg, st[2];
    float lrun = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { ot[0][i] = 0.f; ot[1][i] = 0.f; cneg[i] = 0.f; }
    u32x4 pa[4];

    auto scores = [&](unsigned koff) {
        const char* kb = smem + koff;
        const char* qbase = smem + Q_OFF + wid * (KV_TILE_BYTES / 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = k_row_off + (((2 * ks + hh) ^ k_swz) << 4);
            const bf16x8 q0 = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pa[ks]) : *reinterpret_cast<const bf16x8*>(qbase + off);
            const bf16x8 k0 = (DIAG & 8) ? q0 : *reinterpret_cast<const bf16x8*>(kb + off);
            const bf16x8 k1 = (DIAG & 8) ? q0 : *reinterpret_cast<const bf16x8*>(kb + 32 * 128 + off);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, ks == 0 ? cneg : st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0, ks == 0 ? cneg : st[1], 0, 0, 0);
        }
    };
    auto softmax = [&](int t) {
        if constexpr (!(DIAG & 1)) {
            float a = fmaxf(st[0][0], st[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) a = fmaxf(a, fmaxf(st[0][r], st[1][r]));
            float s0, s1;
            swap32(a, s0, s1);
            const float dm = fmaxf(s0, s1);
            if (t == 0 || !__all(dm <= RESCALE_THR_LOG2)) {
                const float up = ceilf(t == 0 ? dm : fmaxf(dm, 0.f));
#pragma unroll
                for (int i = 0; i < 16; ++i) { st[0][i] -= up; st[1][i] -= up; cneg[i] -= up; }
                if (t > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-up);
                    lrun *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { ot[0][i] *= alpha; ot[1][i] *= alpha; }
                }
            }
        }
        float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float e0 = st[s2 >> 1][8 * (s2 & 1) + 2 * j], e1 = st[s2 >> 1][8 * (s2 & 1) + 2 * j + 1];
                if constexpr (!(DIAG & 1)) {
                    e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
                    if (j & 1) { l2 += e0; l3 += e1; } else { l0 += e0; l1 += e1; }
                }
                pa[s2][j] = pack_bf16x2(e0, e1);
            }
        }
        lrun += (l0 + l1) + (l2 + l3);
    };
    auto matrix = [&](int t) {
        const unsigned slot = (unsigned)(KV_TILE_BYTES * (1 + 2 * (t % NB)));
        const char* vb = smem + slot + KV_TILE_BYTES;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                const bf16x8 vf = (DIAG & 8) ? __builtin_bit_cast(bf16x8, pa[(s2 + db) & 3])
                                             : cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pa[s2]), ot[db], 0, 0, 0);
            }
        if (t + 1 < nkt) scores(slot);
    };

    // prologue
    glds16(ksrc, lds_base + piece * 1024);   // K(0): both halves of the workgroup write the same bytes (uniform load counts)
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j < nkt) dma_bundle(j);
    if (issued >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (issued == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    barrier();
    scores(0);
    if (grp == 1) { wait_bundle(0); barrier(); }
    for (int t = 0; t < nkt; ++t) {
        const bool refill = t >= 1 && t - 1 + NB < nkt && !(DIAG & 2);
        if (grp == 1 && refill) dma_bundle(t - 1 + NB);
        __builtin_amdgcn_sched_barrier(0);
        softmax(t);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) wait_bundle(t);
        barrier();
        if (grp == 0 && refill) dma_bundle(t - 1 + NB);
        __builtin_amdgcn_sched_barrier(0);
        matrix(t);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1 && t + 1 < nkt) wait_bundle(t + 1);
        barrier();
    }
    if (grp == 0) barrier();

    float s0, s1;
    swap32(lrun, s0, s1);
    const float inv = 1.0f / (s0 + s1);
    if (!qvalid) return;
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
            } else {
                u32x2 st2;
                st2[0] = pack_bf16x2(o[0], o[1]);
                st2[1] = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<u32x2*>(p.out + grow * p.ldo + col) = st2;
            }
        }
}
