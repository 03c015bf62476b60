// attn64p.h — attn64p (round 6): the fused head_dim-64 attention forward with SIXTY-FOUR queries per wave — two 32-query blocks
// A and B — at TWO waves per SIMD (256 registers), 256 queries per workgroup, two workgroups per CU.
// Include inside `namespace ditto { namespace {` after attn64v2.h (constants DH, KBLK, KV_TILE_BYTES, RESCALE_THR_LOG2).
// Compile WITHOUT the SLP vectoriser (attention_p.hip: -fno-slp-vectorize): packed v_pk_add_f32 / v_pk_mul_f32 occupy the vector
// pipe for two passes beside the MFMAs and cost 40 more registers here.
//
// Replaces  softmax(q k^T / sqrt(dh)) v  of the self-attention (reference src/components/DiT.py:131-134, head merge + residual
// :137-139) and of nn.MultiheadAttention's cross-attention (:144-148), as attn64v2 does; same contract (q PRE-SCALED by
// scale * log2(e) at pack time, S^T = K Q^T layout with the query on the lane, exp2-domain online softmax with the raise of the
// running maximum deferred by 2^8 and taken in whole octaves).
//
// Why this structure (measurements: tools/probe_attn64p.hip -> profiles/r06_attn_probe.txt).  Rounds 2-5 ran one 32-query block
// per wave at 3 (attn64v2), 2 (attn64v3), 4 (attn64w4) waves per SIMD and two blocks per wave at ONE wave per SIMD (attn64v4): all
// within 10 % of one another.  The knock-outs of this round say why: at head_dim 64 a tile's VECTOR work (64 exponentials at 8
// issue cycles, 64 + 32 + 32 adds / converts / maxima at 4) is as long as its MFMAs (32 x 32 cycles) and the two overlap only by
// about half whatever the wave structure — so what pays is removing work from both pipes, not re-arranging it:
//   * every K / V fragment read from LDS serves TWO MFMAs (one per block): 24 LDS reads per 32 MFMAs instead of 24 per 16, and a
//     workgroup's K / V tile serves 256 queries: half the LDS-DMA pieces, ring writes and barriers per FLOP;
//   * 32 MFMAs per 64 queries x 64 keys, not 40: the row sum is fp32 adds on the lane's own probabilities (attn64v2 spent four
//     all-ones MFMAs per block on it: 20 % of the matrix pipe's time at a point where that pipe is not the idle one);
//   * NO row maximum on the common path: the raise of the running maximum is decided by the tile's row SUMS (see tile_body), which
//     the kernel needs anyway — the maximum (two chains of v_max3_f32 + a lane exchange per block, a sixth of the vector work) is
//     taken on the first tile and on the rare tile whose sums say a raise is due;
//   * the eight K fragments of a tile requested together ahead of the sixteen score MFMAs (left alone hipcc issues read, wait,
//     MFMA, read ... and every MFMA pays an LDS round trip);
//   * the epilogue trades 4-column groups between the two halves of a query (v_permlane32_swap) so that every lane stores 16
//     contiguous bytes per row instead of 8 (guide T21: the store tail of a workgroup is issue-bound).
// The running maximum stays the score chains' initial accumulator (-m in a 16-register block per query block: no per-element
// subtract).  Two waves per SIMD, not one: attn64v4's lone wave issued one vector instruction per 4 cycles with nobody to cover
// its LDS and DMA waits.  Tried on this kernel and NOT kept (profiles/r06_attn_probe.txt): block B's exponentials placed between
// block A's O += V P MFMAs by scheduling-group barriers (equal or slower at 3 spilled registers), Q^T re-read from LDS to free 32
// registers for that (slower), s_setprio around the MFMA clusters (+8 %), 512-query workgroups (slower at Sq = 1024: nothing covers
// a workgroup's prologue and epilogue), a start stagger of the CU's second workgroup (no effect), warm-up loads for the block that
// takes the slot next (-4 % from HBM in isolation, +1 % in the model), v_dot2c_f32_bf16 for the row sum (not faster than two adds).
//
// LDS: a ring of NBUF {K, V} tile pairs (4 x 16 KiB per workgroup).  Tile t + NBUF - 1 is issued at the top of iteration t into the
// slot of tile t - 1 (free: every wave passed the barrier that ended iteration t - 1), and the counted wait in front of the barrier
// that ends iteration t leaves the younger tiles in flight: one barrier per 32 MFMAs per wave.
#pragma once

// this lane's value and lane i ^ 32's (the other 32 keys of the same query): one v_permlane32_swap (vdst lanes 32..63 <-> src lanes
// 0..31) on two COPIES of the value.  As an asm statement with its own wait states: a vector write of a permlane operand must be two
// instructions back, and the builtin handed the same value twice gets ONE register for both operands (which swaps a register's
// halves with themselves: each half then sees only the other half's value — measured: rel-L2 4e-2 in the output).
DITTO_DEV void swap32(float a, float& lo, float& hi) {
    float x = a, y = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    lo = x; hi = y;   // x = {a[0..31], a[0..31]}, y = {a[32..63], a[32..63]}
}
// v_max3_f32 through the builtins (the backend fuses the pair under -fno-honor-nans), NOT an asm statement: its operands are MFMA
// results, and the wait states gfx950 needs between an MFMA's write and a vector read of the register are inserted by the compiler
// for its own instructions only — an asm statement scheduled right behind the last MFMA of a score block reads the accumulator one
// MFMA short (found in attn64q.h's first prologue, round 6).
DITTO_DEV float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

constexpr float SUM_RAISE_THR = 8192.0f;   // 2^13: a lane's 32 probabilities of a tile may sum to this before the tile is redone with a raise

// DIAG (tools/probe_attn64p.hip only; wrong results by design, every knock-out computes on VALID data): bit 0 = no softmax arithmetic
// (P = the packed scores), bit 1 = no K/V DMA after the prologue (every ring slot holds a tile), bit 2 = no barrier / DMA wait in the
// loop (with bit 1), bit 3 = no LDS fragment reads (Q fragments stand in), bit 4 = the exponentials replaced by adds, bit 7 = no MFMAs
// the workgroup's whole job as a device function over a ring of NBUF x 16 KiB of LDS that is free on entry (attn64q.h runs it as
// its exact path)
template <bool RESID, int NBUF, int DIAG>
DITTO_DEV void attn64p_body(const AttnParams& p, char* smem, int tid, int bid) {
    static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");
    constexpr int QWG = 256;                                      // queries per workgroup (p.nqb counts blocks of this size)
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.nqb * p.H * p.B;
    const int id = xcd_remap(bid, nwg);
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
    const int nkt = (p.Skv + KBLK - 1) / KBLK;
    const bool ragged = (p.Skv & (KBLK - 1)) != 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    bf16x8 qf[2][4];   // Q^T B-operand fragments: lane holds Q[query ql of block x][d = 16 ks + 8 hh + 0..7]
    {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16* qp = p.q + ((size_t)b * p.Sq + qrow[x]) * p.ldq + h * DH + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[x][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
        }
    }
    // this lane's two (row, chunk) DMA sources of tile 0 (LDS swizzles applied on the source); tile kt is + kt * 64 rows
    const bf16 *ksrc[2], *vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3), cpos = lane & 7;
        ksrc[i] = p.k + ((size_t)b * p.Skv + row) * p.ldk + h * DH + (cpos ^ ((row >> 1) & 7)) * 8;
        vsrc[i] = p.v + ((size_t)b * p.Skv + row) * p.ldv + h * DH + (cpos ^ (((row >> 1) & 1) << 2)) * 8;
    }
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    auto dma_kv = [&](int kt, int slot) {   // 4 loads per wave
        if (ragged && kt == nkt - 1) {      // rows past Skv are clamped (never read out of bounds), masked in the tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = wid * 2 + i;
                const int row = piece * 8 + (lane >> 3), cpos = lane & 7;
                int key = kt * KBLK + row;
                key = key < p.Skv ? key : p.Skv - 1;
                const int ck = cpos ^ ((row >> 1) & 7), cv = cpos ^ (((row >> 1) & 1) << 2);
                glds16(p.k + ((size_t)b * p.Skv + key) * p.ldk + h * DH + ck * 8,
                       lds_base + (unsigned)(slot * 2 * KV_TILE_BYTES + piece * 1024));
                glds16(p.v + ((size_t)b * p.Skv + key) * p.ldv + h * DH + cv * 8,
                       lds_base + (unsigned)(slot * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            glds16(ksrc[i] + (size_t)kt * kstep, lds_base + (unsigned)(slot * 2 * KV_TILE_BYTES + piece * 1024));
            glds16(vsrc[i] + (size_t)kt * vstep,
                   lds_base + (unsigned)(slot * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
        }
    };
    // wait until at most `groups` younger tile groups (4 loads per wave each) are in flight
    auto wait_groups = [&](int groups) {
        if (groups >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (groups == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    const int k_row_off = ql * 128, k_swz = (ql >> 1) & 7;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int tr_colbyte = (16 * ((lane >> 4) & 1) + 4 * tr_p) * 2;
    const int tr_row0 = 4 * hh + tr_q;
    const int tr_swz = ((tr_q >> 1) & 1) << 6;

    // cneg[x]: every register = -m_run of block x, the score chains' initial accumulator (no per-element subtract)
    f32x16 ot[2][2], cneg[2];
    float lrun[2] = {0.f, 0.f};   // this lane's share (its 32 of a tile's 64 keys) of the row sums
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i) { ot[x][0][i] = 0.f; ot[x][1][i] = 0.f; cneg[x][i] = 0.f; }

    // ---- prologue: tile 0 ALONE first, then tiles 1 .. NBUF-2.  The workgroups of a launch start together (512 resident, three
    //      rounds at C2): with all three tiles of every workgroup requested at once the FIRST tiles queue behind 32 KiB more per
    //      workgroup and nothing computes meanwhile (probe, same process: 121.0 -> 114.7 us warm, 130.4 -> 123.8 from HBM) ----
    dma_kv(0, 0);
    wait_groups(0);
    // The Q fragments are global loads whose first use is INSIDE the tile loop: hipcc then keeps their `s_waitcnt vmcnt(3) .. (0)` in
    // the loop body, where they drain the LDS-DMA pipeline once per tile (rounds 6's first builds ran that way: tools/check_attn_loop.py).
    // A use in front of the loop moves the compiler's wait here, next to the vmcnt(0) above.
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[x][ks]));
    if constexpr (DIAG & 2) {   // (knock-out: every slot holds a tile, no traffic in the loop)
#pragma unroll
        for (int t0 = 1; t0 < NBUF; ++t0) dma_kv(t0 < nkt ? t0 : 0, t0);
        wait_groups(0);
    } else {
#pragma unroll
        for (int t0 = 1; t0 < NBUF - 1; ++t0)
            if (t0 < nkt) dma_kv(t0, t0);
    }
    __syncthreads();

    auto tile_body = [&](int kt, int slot, auto MASKED) {
        const char* kb = smem + slot * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;
        if (!(DIAG & 2) && kt + NBUF - 1 < nkt) {
            int ns = slot + NBUF - 1;
            ns = ns >= NBUF ? ns - NBUF : ns;
            dma_kv(kt + NBUF - 1, ns);
        }

        f32x16 st[2][2];
        u32x4 pa[4], pb[4];
        float lt[2];   // this lane's sums of the tile's probabilities, per block
        // ---- S'^T[key][query] = K Q'^T - m for both blocks: every K fragment feeds two MFMAs.  The eight fragments are requested
        //      together, ahead of the first MFMA (the registers exist: S', P and the V fragments are dead here) ----
        auto scores = [&]() {
            bf16x8 kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                kf[i] = (DIAG & 8) ? qf[i & 1][i >> 1]
                                   : *reinterpret_cast<const bf16x8*>(kb + (i >> 2) * 32 * 128 + k_row_off + (((2 * (i & 3) + hh) ^ k_swz) << 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if constexpr (DIAG & 128) {   // no MFMAs: the scores are the (valid) previous ones, kept alive
                        if (ks == 0 && kt == 0) { st[0][kb2] = cneg[0]; st[1][kb2] = cneg[1]; }
                        asm volatile("" : "+v"(st[0][kb2]), "+v"(st[1][kb2]) : "v"(kf[kb2 * 4 + ks]));
                    } else {
                        st[0][kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2 * 4 + ks], qf[0][ks], ks == 0 ? cneg[0] : st[0][kb2], 0, 0, 0);
                        st[1][kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb2 * 4 + ks], qf[1][ks], ks == 0 ? cneg[1] : st[1][kb2], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(MASKED)::value) {   // ragged last tile only: keys >= Skv never contribute
                const int kbase_idx = kt * KBLK + 4 * hh;
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kbase_idx + kb2 * 32 + (r & 3) + 8 * (r >> 2);
                            if (key >= p.Skv) st[x][kb2][r] = -1e30f;
                        }
            }
        };
        // ---- the exact row maxima relative to the running ones, and the raise: the running maximum moves to (at least) the tile's, in
        //      whole octaves (the factors are powers of two); S', -m, O and l are rescaled together, BEFORE any P of this tile exists and
        //      with the previous tile's P V complete (guide T13 safe order) ----
        auto raise_blk = [&](int x) {
            if constexpr (!(DIAG & 1)) {
                float c0 = max3f(st[x][0][0], st[x][0][1], st[x][0][2]), c1 = max3f(st[x][1][0], st[x][1][1], st[x][1][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) { c0 = max3f(c0, st[x][0][r], st[x][0][r + 1]); c1 = max3f(c1, st[x][1][r], st[x][1][r + 1]); }
                const float a = max3f(c0, c1, fmaxf(st[x][0][15], st[x][1][15]));
                float s0, s1;
                swap32(a, s0, s1);
                const float dm = fmaxf(s0, s1);
                // first tile: the running maximum IS this tile's; later: it only ever rises
                const float up = ceilf(kt == 0 ? dm : fmaxf(dm, 0.f));
#pragma unroll
                for (int i = 0; i < 16; ++i) { st[x][0][i] -= up; st[x][1][i] -= up; cneg[x][i] -= up; }
                if (kt > 0) {   // (first tile: O and l are zero, and 2^-up may be infinite)
                    const float alpha = __builtin_amdgcn_exp2f(-up);
                    lrun[x] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { ot[x][0][i] *= alpha; ot[x][1][i] *= alpha; }
                }
            }
        };
        auto raise_exact = [&]() { raise_blk(0); raise_blk(1); };
        // ---- P = exp2(S') as packed bf16 pairs (the B operands of O += V P); lt = this lane's fp32 sum of the probabilities
        //      (v_dot2c_f32_bf16 on the packed pairs was tried for it: not faster than the two adds it replaces, and hipcc 7.2 miscompiles
        //      the builtin on two accumulator chains) ----
        // (tools/probe_coissue.hip, profiles/r06_coissue.txt: a vector instruction that CONSUMES a transcendental's result makes no
        // progress while the other wave of the SIMD streams MFMAs back to back — pure v_exp / v_add / v_cvt streams lose 25 % there, a
        // stream in which adds or converts read v_exp results waits out the partner's whole burst.  That is why the softmax and the
        // MFMAs of two waves add up rather than overlap, whatever the wave structure.  Putting all 64 exponentials first, as one pure
        // transcendental stream, and the 96 consumers behind it measured SLOWER in this kernel — 121 against 114 us — and was not kept.)
        auto exp_all = [&]() {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float l0 = 0.f, l1 = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    u32x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float e0 = st[x][s2 >> 1][8 * (s2 & 1) + 2 * j], e1 = st[x][s2 >> 1][8 * (s2 & 1) + 2 * j + 1];
                        if constexpr (!(DIAG & 1)) {
                            if constexpr (DIAG & 16) { e0 += 1.0f; e1 += 1.0f; }
                            else { e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); }
                            l0 += e0; l1 += e1;
                        }
                        pk[j] = pack_bf16x2(e0, e1);
                    }
                    if (x == 0) pa[s2] = pk; else pb[s2] = pk;
                }
                lt[x] = l0 + l1;
            }
        };
        // The raise is decided by the SUMS, not by a row maximum taken on every tile (16 v_max3_f32 + a lane exchange per block and
        // tile: a sixth of the vector work): the probabilities are computed against the running maximum as it stands, and only if
        // some lane's 32 of them sum to more than 2^13 (so: every P of the tile <= 2^13, exact in bf16's exponent range, fp32
        // accumulation) — or to inf / NaN — the tile is redone the exact way: scores again (its K tile is still in LDS), the true row
        // maxima, the raise, the exponentials.  Wave-uniform and rare (a key that beats the running maximum by more than 2^8 .. 2^13);
        // the first tile always takes the exact path (it establishes the running maximum).
        scores();
        if (kt == 0) raise_exact();
        exp_all();
        if constexpr (!(DIAG & 1)) {
            if (kt > 0 && !__all(lt[0] <= SUM_RAISE_THR && lt[1] <= SUM_RAISE_THR)) {
                scores();
                raise_exact();
                exp_all();
            }
        }
        lrun[0] += lt[0];
        lrun[1] += lt[1];
        // ---- O^T[d][query] += V^T[d][key] P^T[key][query] ----
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
                const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
                const bf16x8 vf = (DIAG & 8) ? qf[db][s2]
                                             : cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                                                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
                if constexpr (DIAG & 128) {
                    asm volatile("" : "+v"(ot[0][db]), "+v"(ot[1][db]) : "v"(vf), "v"(pa[s2]), "v"(pb[s2]));
                } else {
                    ot[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pa[s2]), ot[0][db], 0, 0, 0);
                    ot[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pb[s2]), ot[1][db], 0, 0, 0);
                }
            }

        // ---- tile kt+1 landed (this wave's pieces), then for everyone; everyone is done with this tile's slot ----
        if constexpr (!(DIAG & 4)) {
            const int last = kt + NBUF - 1 < nkt ? kt + NBUF - 1 : nkt - 1;   // youngest tile issued
            if constexpr (!(DIAG & 2)) wait_groups(last - (kt + 1));
            __syncthreads();
        }
    };
    {
        const int nfull = ragged ? nkt - 1 : nkt;
        int slot = 0;
        for (int kt = 0; kt < nfull; ++kt) {
            tile_body(kt, slot, std::false_type{});
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
        if (ragged) tile_body(nkt - 1, slot, std::true_type{});
    }

    // ---- epilogue: normalise; lane (query ql, half hh) owns d = 32 db + 8 g + 4 hh + 0..3 of its two queries.  The two halves of a
    //      query (lanes q and q + 32) trade 4-column groups so that every lane owns 8 CONSECUTIVE head columns — lane q: 32 db + 16 k
    //      + 0..7, lane q + 32: 32 db + 16 k + 8..15 — and a row's 16 bytes (bf16) leave in one store instead of two 8-byte ones ----
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float s0, s1;
        swap32(lrun[x], s0, s1);
        const float inv = 1.0f / (s0 + s1);
        const size_t grow = (size_t)b * p.Sq + qrow[x];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float lo[4], hi[4];   // groups g = 2 k2 (columns +0..3 / +4..7 by half) and g = 2 k2 + 1 (+8..11 / +12..15)
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = ot[x][db][8 * k2 + e] * inv; hi[e] = ot[x][db][8 * k2 + 4 + e] * inv; }
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo[e]), "+v"(hi[e]));
                // lanes 0..31: lo = own columns +0..3, hi = the partner's +4..7; lanes 32..63: lo = the partner's +8..11, hi = own +12..15
                if (!qvalid[x]) continue;
                const int col = h * DH + 32 * db + 16 * k2 + 8 * hh;
                if constexpr (RESID) {   // x = attn_out + residual (src/components/DiT.py:139), on the bf16 stream or the fp32 one
                    if (p.resid_bf16) {
                        const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(p.resid_in) + grow * p.ldr + col);
                        u32x4 o4;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            o4[e] = pack_bf16x2(lo[2 * e] + bf16_lo(w[e]), lo[2 * e + 1] + bf16_hi(w[e]));
                            o4[2 + e] = pack_bf16x2(hi[2 * e] + bf16_lo(w[2 + e]), hi[2 * e + 1] + bf16_hi(w[2 + e]));
                        }
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.resid) + grow * p.ldr + col) = o4;
                    } else {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(p.resid_in + grow * p.ldr + col);
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(p.resid_in + grow * p.ldr + col + 4);
                        f32x4 o0, o1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o0[e] = r0[e] + lo[e]; o1[e] = r1[e] + hi[e]; }
                        *reinterpret_cast<f32x4*>(p.resid + grow * p.ldr + col) = o0;
                        *reinterpret_cast<f32x4*>(p.resid + grow * p.ldr + col + 4) = o1;
                    }
                } else {
                    u32x4 o4;
                    o4[0] = pack_bf16x2(lo[0], lo[1]); o4[1] = pack_bf16x2(lo[2], lo[3]);
                    o4[2] = pack_bf16x2(hi[0], hi[1]); o4[3] = pack_bf16x2(hi[2], hi[3]);
                    *reinterpret_cast<u32x4*>(p.out + grow * p.ldo + col) = o4;
                }
            }
    }
}

template <bool RESID, int NBUF = 4, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void attn64p_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * KV_TILE_BYTES];  // [slot][K|V]
    attn64p_body<RESID, NBUF, DIAG>(p, smem, threadIdx.x, blockIdx.x);
}
