// attn64q.h — attn64q (round 6): attn64p's wave (64 queries = blocks A and B, two waves per SIMD, 256-query workgroups) as ONE
// software-pipelined instruction stream in which every MFMA is followed by one "pair step" of the softmax of ANOTHER block / tile.
// Include inside `namespace ditto { namespace {` after attn64p.h.  Compile without the SLP vectoriser.
//
// Why (tools/probe_coissue.hip, profiles/r06_coissue.txt).  On gfx950 a vector instruction that consumes a transcendental's result
// makes NO progress while the other wave of its SIMD has MFMAs in flight; pure streams (exponentials, adds, converts on their own)
// lose only 25 % there.  So with the usual phase structure — all of a tile's score MFMAs, then its softmax, then its P V MFMAs — the
// two waves of a SIMD exclude each other: wave X's MFMA phase blocks wave Y's softmax, and the pipes' times ADD (attn64p: 2 100
// cycles per 64 queries x 64 keys per SIMD for 1 024 of MFMA and ~1 150 of vector issue).  Inside ONE wave the two kinds do
// overlap: one MFMA followed by {2 v_exp, 2 v_add, 1 v_cvt_pk} costs 47.5 cycles at one wave per SIMD and 42.6 per SIMD with two
// such waves (against 32 for the MFMA alone), i.e. ~1 400 cycles per tile.  This kernel is that stream:
//
//   slot (one MFMA each)      matrix pipe                          vector pipe (one pair step = 2 exp, 2 add, 1 cvt_pk)
//    0 ..  7                  S_A(t)   = K(t) Q_A^T                 P_B(t-1), second half  (pairs 8..15 of S_B(t-1))
//    8 .. 15                  O_B     += V(t-1) P_B(t-1)            P_A(t),   first half
//   16 .. 23                  S_B(t)   = K(t) Q_B^T                 P_A(t),   second half
//   24 .. 31                  O_A     += V(t) P_A(t)                P_B(t),   first half
//
// (the half-tile skew of round 3's attn64v4, at two waves per SIMD and with a third fewer vector instructions).  LDS fragments are
// requested QD slots ahead of the MFMA that takes them, through a queue of registers that runs on over the loop's back edge (the
// per-tile barrier sits at slot 32 - QD, so that the first K fragments of the next tile are requested behind it).  HOLD = 1 keeps a
// tile's 8 K fragments in registers from S_A's slot to S_B's (24 LDS fragment reads per tile instead of 32; the plain form), HOLD
// = 0 reads every fragment per use (the residual form: its epilogue needs the registers; HOLD = 2, V fragments held as well,
// spills one fragment there, and one scratch reload in the loop counts on vmcnt with the LDS-DMA loads — every counted wait
// becomes a drain: 165 against 117 us).  Measured (profiles/r06_attn64q.txt): C2 B = 32 in isolation 104-106 us against 117-122
// (attn64p) and 125 (attn64v2); in the model self 110.6 / cross 104.0 against 117.0 / 116.2 (attn64p) on one box.
//
// WHAT BOUNDS IT.  v_exp_f32 is quarter rate (16 cycles per wave64 instruction): a pair step is 2 x 16 + 3 x 4 = 44 cycles of vector
// issue against the MFMA's 32, so the stream is bound by the vector pipe, 73 % of it transcendentals (measured 42.6 per SIMD).  Tried
// on top (profiles/r06_coissue.txt, r06_attn64q.txt): the row sums on the matrix pipe (8 all-ones MFMAs per tile, pair step of
// {exp, exp, cvt} = 35 cycles, 40 slots): 111 against 105 us — the other per-slot costs scale with the MFMA count; v_dot2(c)_f32_bf16
// for the sums: waits for the matrix pipe (49 cycles per step); consumers one step behind their exponentials: -3 % in the
// microbenchmark, not built.
//
// ASM PAIR STEPS AND THE MFMA HAZARD.  The pair step of the steady loop is ONE asm statement (order = schedule; the compiler can
// neither sink the adds out of the block nor pack them).  gfx950 needs software wait states between an MFMA's write of a register
// and a vector instruction's read of it; hipcc inserts them for its own instructions and NOT for an asm statement's.  In the loop
// every slot is pinned by sched_barriers and an asm step reads a score block whose last MFMA lies >= 4 slots (128 cycles) back.
// The prologue and the drain, which the compiler schedules freely, use the same step written in builtins (pair_step_c): the first
// version's asm steps there read their accumulators one MFMA short (rel-L2 2e-2, block A only).
//
// OPTIMISTIC SOFTMAX.  There is no running maximum: P = exp2(S) as it leaves the MFMA (q is pre-scaled: log2 units), O and l
// accumulate unshifted and O / l at the end is the softmax — the same arithmetic as the shifted form as long as nothing leaves
// fp32's range, i.e. for logits within +-88 of zero.  That frees the 32 registers of the -m accumulator blocks and every maximum,
// raise and rescale.  Outside the range a row's l comes out 0, inf or NaN: every wave checks its rows at the end, the workgroup
// agrees through one LDS word, and if any row failed the WHOLE workgroup redoes its block with the exact tile loop (attn64p's:
// running maximum, deferred raise) before anything is stored.  Fast path and exact path differ only by fp32 rounding.
//
// Contract: as attn64p, plus Skv >= 65 (at least two tiles; the last may be partial: its iteration is a second copy of the loop body
// whose score chains start from -inf on the missing key rows); the launcher sends shorter key sequences to attn64p.
#pragma once

constexpr int Q_QD = 4;   // fragments are requested this many slots ahead of their MFMA
#ifndef Q_HOLD_DEFAULT
#define Q_HOLD_DEFAULT 1
#endif
constexpr int Q_HOLD = Q_HOLD_DEFAULT;   // 0: every fragment read per use; 1: K fragments held for block B; 2: K and V fragments held
constexpr int Q_QN = 8;   // queue registers (a power of two that divides 32, > Q_QD)

#ifndef DITTO_STATIC_FOR
#define DITTO_STATIC_FOR
template <class F, int... I>
DITTO_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
DITTO_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#endif

template <bool RESID, int DIAG = 0, int QD = Q_QD, bool WRAP = true, int HOLD = Q_HOLD, bool RAGGED = false>
__global__ __launch_bounds__(256, 2) void attn64q_kernel(AttnParams p) {
    constexpr int NBUF = 4, QWG = 256;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * KV_TILE_BYTES];  // [slot][K|V]
    __shared__ int redo;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) redo = 0;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int nkt = (p.Skv + KBLK - 1) / KBLK;
    // RAGGED instantiation: the last tile may be partial — its DMA rows are clamped, its scores start from -inf.  (Its own instantiation:
    // the second copy of the loop body raises the register pressure in front of the loop, and in the plain form that put a scratch
    // reload there whose vmcnt wait hipcc then keeps INSIDE the loop — a drain of the LDS-DMA pipeline per tile, 145 against 102 us.)
    const bool ragged = RAGGED && (p.Skv % KBLK) != 0;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    bf16x8 qf[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const bf16* qp = p.q + ((size_t)b * p.Sq + qrow[x]) * p.ldq + h * DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[x][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
    const bf16 *ksrc[2], *vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 8 + (lane >> 3), cpos = lane & 7;
        ksrc[i] = p.k + ((size_t)b * p.Skv + row) * p.ldk + h * DH + (cpos ^ ((row >> 1) & 7)) * 8;
        vsrc[i] = p.v + ((size_t)b * p.Skv + row) * p.ldv + h * DH + (cpos ^ (((row >> 1) & 1) << 2)) * 8;
    }
    const size_t kstep = (size_t)KBLK * p.ldk, vstep = (size_t)KBLK * p.ldv;
    auto dma_kv = [&](int kt, int slot) {   // 4 loads per wave
        if constexpr (DIAG & 2) return;
        if (ragged && kt == nkt - 1) {      // rows past Skv are clamped (never read out of bounds; finite values under P = 0)
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
            glds16(vsrc[i] + (size_t)kt * vstep, lds_base + (unsigned)(slot * 2 * KV_TILE_BYTES + KV_TILE_BYTES + piece * 1024));
        }
    };
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

    f32x16 ot[2][2], st[2][2];
    u32x4 pp[2][4];
    float lsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i) { ot[x][0][i] = 0.f; ot[x][1][i] = 0.f; }

    // fragment of MFMA slot m of an iteration whose K / V tile sits at kvb (this tile) and whose previous tile's V at vprev
    auto kfrag = [&](const char* kb, int kb2, int ks) {
        return *reinterpret_cast<const bf16x8*>(kb + kb2 * 32 * 128 + k_row_off + (((2 * ks + hh) ^ k_swz) << 4));
    };
    auto vfrag = [&](const char* vb, int s2, int db) {
        const int colb = (tr_colbyte + 64 * db) ^ tr_swz;
        const char* a0 = vb + (16 * s2 + tr_row0) * 128 + colb;
        return cat4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0)),
                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 8 * 128)));
    };
    // one pair step: the probabilities of elements 2 i, 2 i + 1 of S_x[kb2]
    auto pair_step = [&](auto X_, auto KB2_, auto I_) {
        constexpr int x = decltype(X_)::value, kb2 = decltype(KB2_)::value, i = decltype(I_)::value;
        if constexpr (DIAG & 1) {
            pp[x][2 * kb2 + (i >> 2)][i & 3] = pack_bf16x2(st[x][kb2][2 * i], st[x][kb2][2 * i + 1]);
        } else {
            // one asm statement: the order is the schedule (an exponential's result is not consumed by the instruction right behind it), and
            // the compiler can neither sink the adds out of the loop body's first block nor pack them
            float e0, e1;
            unsigned w;
            if constexpr (DIAG & 16)
                asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\ts_nop 0\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                             : "=&v"(e0), "=&v"(e1), "=v"(w) : "v"(st[x][kb2][2 * i]), "v"(st[x][kb2][2 * i + 1]));
            else
            asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %3, %3, %0\n\tv_add_f32 %4, %4, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                         : "=&v"(e0), "=&v"(e1), "=v"(w), "+v"(lsum[x][0]), "+v"(lsum[x][1])
                         : "v"(st[x][kb2][2 * i]), "v"(st[x][kb2][2 * i + 1]));
            pp[x][2 * kb2 + (i >> 2)][i & 3] = w;
        }
    };

    // the same step in builtins, for the prologue and the drain: there the compiler schedules freely, and it must SEE the reads of MFMA
    // results (an MFMA's write followed by a vector read of the register needs software wait states on gfx950: the compiler inserts
    // them for its own instructions, not for an asm statement's — a pair step scheduled right behind the last MFMA of its score block
    // read the accumulator one MFMA short).  In the steady loop the slots are pinned and every asm step reads a block whose last MFMA
    // is at least four MFMA slots (128 cycles) back.
    auto pair_step_c = [&](auto X_, auto KB2_, auto I_) {
        constexpr int x = decltype(X_)::value, kb2 = decltype(KB2_)::value, i = decltype(I_)::value;
        float e0 = st[x][kb2][2 * i], e1 = st[x][kb2][2 * i + 1];
        if constexpr (!(DIAG & 1)) {
            e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
            if constexpr (!(DIAG & 16)) { lsum[x][0] += e0; lsum[x][1] += e1; }
        }
        pp[x][2 * kb2 + (i >> 2)][i & 3] = pack_bf16x2(e0, e1);
    };

    // HOLD: a tile's 8 K fragments stay in registers from S_A's slot to S_B's (16 slots on), its 8 V fragments from O_A's slot to O_B's
    // (16 slots on, across the loop's back edge): 16 LDS fragment reads per tile instead of 32
    bf16x8 kh[HOLD ? 8 : 1], vh[HOLD == 2 ? 8 : 1];
    // ---- prologue: tiles 0, 1, 2 in flight (tile 0 alone first); S_A(0), S_B(0); P_A(0); O_A += V(0) P_A(0); first half of P_B(0) ----
    dma_kv(0, 0);
    wait_groups(0);
    dma_kv(1, 1);
    if (2 < nkt) dma_kv(2, 2);
    __syncthreads();
    {
        const char* kb = smem;
        const char* vb = smem + KV_TILE_BYTES;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = kfrag(kb, kb2, ks);
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                st[0][kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], ks == 0 ? z : st[0][kb2], 0, 0, 0);
                st[1][kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], ks == 0 ? z : st[1][kb2], 0, 0, 0);
            }
        static_for<8>([&](auto I) { pair_step_c(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, I); });
        static_for<8>([&](auto I) { pair_step_c(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, I); });
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 vf = vfrag(vb, s2, db);
                if constexpr (HOLD == 2) vh[s2 * 2 + db] = vf;
                ot[0][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pp[0][s2]), ot[0][db], 0, 0, 0);
            }
        static_for<8>([&](auto I) { pair_step_c(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, I); });
    }
    // tile 1 landed for everyone (the groups of tile 2 may stay in flight)
    wait_groups(nkt > 2 ? 1 : 0);
    __syncthreads();
    if (3 < nkt) dma_kv(3, 3);

    // ---- steady state: iteration t = 1 .. nkt-1 ----
    int slot = 1;
    bf16x8 fq[Q_QN];     // the fragment queue: slot m's fragment in fq[m % Q_QN], requested QD slots ahead
    {
        const char* kb1 = smem + 2 * KV_TILE_BYTES;
        static_for<QD>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            if constexpr (HOLD) kh[m] = kfrag(kb1, m >> 2, m & 3);
            else fq[m % Q_QN] = kfrag(kb1, m >> 2, m & 3);
        });
    }
    // The residual form on the bf16 stream fetches its rows of the stream by LDS-DMA at the top of the LAST tile, into the two ring
    // slots that no tile will use any more (4 waves x 64 rows x 128 B = 32 KiB = two slots; every wave its own 8 KiB, chunk c of row
    // r at position c ^ (r & 7)): the epilogue then reads data that have had a tile's time to arrive instead of issuing global
    // loads behind the last MFMA, and no register lives across the loop for it (as prefetched VALUES the compiler spilled them to
    // scratch on the spot).  The last tile's barrier waits with vmcnt(0) anyway.  A workgroup that falls back discards them.
    auto iteration = [&](auto MASKED_, const int t) {
        constexpr bool MASKED = decltype(MASKED_)::value;
        // the partial last tile: its score chains start from -inf on the key rows past Skv (P = exp2(-inf) = 0: no instruction in the
        // pair steps, 32 registers that exist in this copy of the iteration only)
        f32x16 zmask[MASKED ? 2 : 1];
        if constexpr (MASKED) {
            const int rem = p.Skv - (nkt - 1) * KBLK;
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                for (int i = 0; i < 16; ++i) zmask[kb2][i] = (32 * kb2 + 8 * (i >> 2) + 4 * hh + (i & 3)) < rem ? 0.f : -__builtin_inff();
        }
        if constexpr (RESID && !(DIAG & 31)) {
            if (t == nkt - 1 && p.resid_bf16) {
                const int rs = (slot + (wid < 2 ? 1 : 2)) & (NBUF - 1);
                const unsigned rdst = lds_base + (unsigned)(rs * 2 * KV_TILE_BYTES + (wid & 1) * KV_TILE_BYTES);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 8 * i + (lane >> 3);
                    int qr = qb * QWG + wid * 64 + row;
                    qr = qr < p.Sq ? qr : p.Sq - 1;
                    glds16(reinterpret_cast<const bf16*>(p.resid_in) + ((size_t)b * p.Sq + qr) * p.ldr + h * DH + (((lane & 7) ^ (row & 7)) << 3),
                           rdst + (unsigned)(i * 1024));
                }
            }
        }
        const char* kb = smem + slot * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;
        const char* vprev = smem + (slot == 0 ? NBUF - 1 : slot - 1) * 2 * KV_TILE_BYTES + KV_TILE_BYTES;
        const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
        const char* kbn = smem + nslot * 2 * KV_TILE_BYTES;   // K(t+1): the fragment queue runs on into the next iteration's first slots
        auto fetch = [&](auto M_) {
            constexpr int m = decltype(M_)::value, mm = m & 31, j = mm & 7;
            if constexpr (HOLD == 2) {
                if constexpr (m >= 32) kh[j] = kfrag(kbn, j >> 2, j & 3);
                else if constexpr (mm < 8) kh[j] = kfrag(kb, j >> 2, j & 3);
                else if constexpr (mm >= 24) vh[j] = vfrag(vb, j >> 1, j & 1);
            } else if constexpr (HOLD == 1) {      // V fragments through the queue, K fragments held
                if constexpr (m >= 32) kh[j] = kfrag(kbn, j >> 2, j & 3);
                else if constexpr (mm < 8) kh[j] = kfrag(kb, j >> 2, j & 3);
                else if constexpr (mm >= 24) fq[m % Q_QN] = vfrag(vb, j >> 1, j & 1);
                else if constexpr (mm >= 8 && mm < 16) fq[m % Q_QN] = vfrag(vprev, j >> 1, j & 1);
            } else if constexpr (m >= 32) fq[m % Q_QN] = kfrag(kbn, j >> 2, j & 3);
            else if constexpr (mm < 8) fq[m % Q_QN] = kfrag(kb, j >> 2, j & 3);
            else if constexpr (mm < 16) fq[m % Q_QN] = vfrag(vprev, j >> 1, j & 1);
            else if constexpr (mm < 24) fq[m % Q_QN] = kfrag(kb, j >> 2, j & 3);
            else fq[m % Q_QN] = vfrag(vb, j >> 1, j & 1);
        };
        if constexpr (!WRAP) static_for<QD>([&](auto M) { fetch(M); });
        static_for<32>([&](auto M_) {
            constexpr int m = decltype(M_)::value, j = m & 7;
            if constexpr (m + QD < 32) fetch(std::integral_constant<int, m + QD>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m == 32 - QD) {
                // tile t+1 landed (this wave's pieces; the groups of tiles t+2, t+3 may stay in flight), then for everyone; everyone is
                // past block 2 of this iteration, so the slot of tile t-1 is free for tile t+3
                const int last = t + 2 < nkt ? t + 2 : nkt - 1;
                wait_groups(last - (t + 1));
                if constexpr (!(DIAG & 4)) __syncthreads();
                if (t + 3 < nkt) dma_kv(t + 3, slot == 0 ? NBUF - 1 : slot - 1);
            }
            if constexpr (WRAP && m + QD >= 32) fetch(std::integral_constant<int, m + QD>{});   // (past the last tile: a read of a dead slot)
            bf16x8 f;
            if constexpr (HOLD == 2) { if constexpr ((m & 8) != 0) f = vh[j]; else f = kh[j]; }
            else if constexpr (HOLD == 1) { if constexpr ((m & 8) != 0) f = fq[m % Q_QN]; else f = kh[j]; }
            else f = fq[m % Q_QN];
            if constexpr (m < 8) {
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                st[0][j >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[0][j & 3], (j & 3) == 0 ? (MASKED ? zmask[MASKED ? (j >> 2) : 0] : z) : st[0][j >> 2], 0, 0, 0);

                __builtin_amdgcn_sched_barrier(0);      // the MFMA first: the pair step issues under it
                pair_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, j>{});
            } else if constexpr (m < 16) {
                ot[1][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, __builtin_bit_cast(bf16x8, pp[1][j >> 1]), ot[1][j & 1], 0, 0, 0);

                __builtin_amdgcn_sched_barrier(0);      // the MFMA first: the pair step issues under it
                pair_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, j>{});
            } else if constexpr (m < 24) {
                f32x16 z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                st[1][j >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[1][j & 3], (j & 3) == 0 ? (MASKED ? zmask[MASKED ? (j >> 2) : 0] : z) : st[1][j >> 2], 0, 0, 0);

                __builtin_amdgcn_sched_barrier(0);      // the MFMA first: the pair step issues under it
                pair_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, j>{});
            } else {
                ot[0][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, __builtin_bit_cast(bf16x8, pp[0][j >> 1]), ot[0][j & 1], 0, 0, 0);

                __builtin_amdgcn_sched_barrier(0);      // the MFMA first: the pair step issues under it
                pair_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, j>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        slot = nslot;
    };
    for (int t = 1; t < nkt - (ragged ? 1 : 0); ++t) iteration(std::false_type{}, t);
    if constexpr (RAGGED) { if (ragged) iteration(std::true_type{}, nkt - 1); }
    // ---- drain: second half of P_B(last), O_B += V(last) P_B(last) ----
    {
        const int ls = slot == 0 ? NBUF - 1 : slot - 1;
        const char* vb = smem + ls * 2 * KV_TILE_BYTES + KV_TILE_BYTES;
        static_for<8>([&](auto I) { pair_step_c(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, I); });
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db)
            {
                bf16x8 vf;
                if constexpr (HOLD == 2) vf = vh[(s2 * 2 + db) & (HOLD == 2 ? 7 : 0)]; else vf = vfrag(vb, s2, db);
                ot[1][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pp[1][s2]), ot[1][db], 0, 0, 0);
            }
    }

    // ---- did every row stay inside fp32's comfortable range?  l in [2^-100, 2^100]: no probability overflowed, and every one that
    // matters (2^-26 of the row's sum) was a normal number.  Otherwise (or NaN) the workgroup starts over on the exact path. ----
    // (the row indices pass through an asm statement here so that none of the epilogue's address arithmetic is hoisted above the loop
    // and kept in registers across it: one spilled register in the loop costs a scratch reload per tile, which counts on vmcnt with
    // the LDS-DMA loads and turns the counted waits into full drains — measured 165 against 117 us)
#pragma unroll
    for (int x = 0; x < 2; ++x) asm volatile("" : "+v"(qrow[x]));
    float linv[2];
    {
        bool bad = false;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float s0, s1;
            swap32(lsum[x][0] + lsum[x][1], s0, s1);
            const float l = s0 + s1;
            // on the bits (positive floats order like their bit patterns; zero, negatives, inf and NaN fall outside): this file is
            // compiled with -fno-honor-nans, a float comparison may be folded into one that NaN passes
            bad |= (__builtin_bit_cast(unsigned, l) - 0x0D800000u) > (0x71800000u - 0x0D800000u);   // 2^-100 .. 2^100
            linv[x] = 1.0f / l;
        }
        if constexpr (!(DIAG & 31)) {
            if (__any(bad) && lane == 0) redo = 1;
            __syncthreads();
            if (redo) {
                attn64p_body<RESID, NBUF, 0>(p, smem, threadIdx.x, blockIdx.x);
                return;
            }
        }
    }
    // ---- epilogue (attn64p's wide form) ----
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float inv = linv[x];
        const size_t grow = (size_t)b * p.Sq + qrow[x];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float lo[4], hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = ot[x][db][8 * k2 + e] * inv; hi[e] = ot[x][db][8 * k2 + 4 + e] * inv; }
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo[e]), "+v"(hi[e]));
                if (!qvalid[x]) continue;
                const int col = h * DH + 32 * db + 16 * k2 + 8 * hh;
                if constexpr (RESID) {
                    if (p.resid_bf16) {
                        const int rs = (slot + (wid < 2 ? 0 : 1)) & (NBUF - 1);     // (slot has moved on by one since the last tile's top)
                        const int rr = 32 * x + ql, ch = 4 * db + 2 * k2 + hh;
                        const u32x4 w = *reinterpret_cast<const u32x4*>(smem + rs * 2 * KV_TILE_BYTES + (wid & 1) * KV_TILE_BYTES + rr * 128 + ((ch ^ (rr & 7)) << 4));
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
