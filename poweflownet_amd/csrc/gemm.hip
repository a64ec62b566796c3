// fp32 MFMA weight-gradient GEMM (dW = dY^T X) of the hot path (gfx950).  The forward / input-gradient GEMM lives in gemm_nt.hip.
//
// Carries what the reference's autograd runs as one `mm` per nn.Linear weight (EdgeAggregation.edge_aggr
// networks/MPN.py:17-21, TAGConv.lins, mask_embd :491-495), restructured to per-NODE operands:
//   dW[i][j] = sum_m A[m][i] * B[m][j]     A = gradient of the layer output (M x na), B = layer input (M x nb),
// M = nodes (1e4..1e6) is the REDUCTION dimension, na, nb <= a few hundred.  Exact fp32 on v_mfma_f32_32x32x2_f32.
//
// ONE launch serves every weight of the network (model.hip defers all weight gradients to the end of the backward
// pass): a task = one 128 x 128 tile of one weight x one contiguous row range, a workgroup = one task, its 8 waves = 4
// row groups x 2 column halves (two waves per SIMD).  A wave holds a 128 x 64 piece of the tile as 4 x 2 interleaved
// 32 x 32 accumulators (128 registers) and streams its own rows straight from global memory:
//   * both operands are row-major with the reduction index as the row -- exactly the MFMA operand order: at step s lane
//     (c = lane & 31, kh = lane >> 5) supplies row m + 2s + kh;
//   * a lane loads FOUR adjacent columns (16 bytes) of A and TWO (8 bytes) of B, so a wave load covers 2 rows x 512 / 256
//     contiguous bytes and the pair feeds 8 MFMAs (tile (ta, tb) = rows 4i + ta, columns 64 half + 2j + tb of the weight):
//     2 vector-memory instructions per 512 MFMA cycles, no LDS staging, no barrier in the loop, hand double-buffered
//     batches of 8 rows; the two column halves of a row group read the same A rows (the second one hits L1);
//   * operands at most 32 columns wide (the 4-wide node features / outputs) take ONE tile with lane c = column c (a
//     narrow B has no column halves: the 8 waves are 8 row groups);
//   * H = 129 = 128 + 1: the odd row / column and the bias gradient (a virtual ones- or rowscale-column of B) never get a
//     tile -- they are VALU dot products off the fragments the wave already holds.
// The row groups take the 8-row batches of the block's range round-robin (one DRAM stream per operand), are summed by a
// fixed LDS tree, and a task split over several blocks publishes partials that tn_combine_kernel sums in block order
// (deterministic, no float atomics) into the nn.Linear gradient layout.
//
// History: v2 gave every wave a 64 x 64 quadrant (2 x 2 tiles, 8-byte loads, quadrants sharing rows through L1) and ran
// one launch + one combine per layer on a side stream: 33-50 TF.  A 128 x 128 tile per wave (256 accumulators in the
// unified VGPR/AGPR file, one wave per SIMD) compiled to 611 spills: VALU cannot read AGPRs, and the cross-wave sum needs
// every accumulator in a VGPR.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T3_THREADS = 512;            // 8 waves = two per SIMD
constexpr int T3_WAVES = 8;
constexpr int T3_R = 8;                    // ring depth: steps (row pairs) in flight per wave
constexpr int T3_FLUSH = 64;                // narrow shapes: ring turns (of T3_R steps = 16 rows) between two-level flushes
constexpr int T3_ROWS = 8;                 // row ranges of blocks are multiples of T3_ROWS x T3_WAVES rows
constexpr int T3_MAX_PAIRS = 28;            // standard.json needs 26 (kernel arguments: 28 x 72 + 56 x 24 bytes < 4 KB)
constexpr int T3_MAX_TASKS = 56;
constexpr int T3_LDS_BYTES = 4 * (4 * 2 * 4 + 4) * 64 * 16;   // the first tree round of the largest shape: 2 row groups x 2 halves x 36 quads
enum { TNF_XCOL = 1, TNF_BIAS = 2, TNF_XROW = 4 };

// quads (float4 per lane) of ONE WAVE's partial: TA*TB tiles x 4 register groups, then 4 quads of VALU extras.  TA = 4 / 1
// (wide / narrow A), TB = 2 / 1 (wide B: the wave's half of the 128 columns / narrow B)
__host__ __device__ constexpr int t3_quads(int TA, int TB) { return TA * TB * 4 + 4; }

struct T3Task {
    short pair, ti, tj, flags;
    short wa, wb;          // 1: the operand is wide (a lane holds 4 adjacent columns); 0: narrow (lane c = column c)
    short gsize, gidx;     // tasks that share an operand form a group: its size (set on every member) and this task's index in it
    int nsplit;            // row splits of this task (equal for the members of a group)
    int block0;            // first workgroup id of the GROUP
    int part0;             // float4 offset of this task's first partial in the partial buffer
};
struct T3Args {
    TnPair pair[T3_MAX_PAIRS];
    T3Task task[T3_MAX_TASKS];
    int ntasks, M, nblocks, stamp_want;
    float4* partial;
    const int* stamp;      // optional guard word (model.hip WS_STAMP_*): if *stamp != stamp_want every gradient is written as NaN
};
__device__ __forceinline__ bool t3_bad(const T3Args& a) { return a.stamp != nullptr && *a.stamp != a.stamp_want; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W>
struct T3Frag {   // one operand of one MFMA step: 4 (A) / 2 (B) adjacent columns of a wide operand, 1 column of a narrow one
    typedef typename std::conditional<W == 4, f32x4, typename std::conditional<W == 2, f32x2, float>::type>::type type;
};
__device__ __forceinline__ float frag_at(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ float frag_at(const f32x2& v, int i) { return v[i]; }
__device__ __forceinline__ float frag_at(const float& v, int) { return v; }
__device__ __forceinline__ void frag_zero(f32x4& v) { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void frag_zero(f32x2& v) { v = f32x2{0.f, 0.f}; }
__device__ __forceinline__ void frag_zero(float& v) { v = 0.f; }

// Scatter one float4 quad of a wave partial into the nn.Linear gradient layout.  Quad Q < 4 TA TB = register group g = Q & 3 of
// tile (ta, tb) = ((Q >> 2) / TB, (Q >> 2) % TB): element e is accumulator row i = e + 8 g + 4 kh, column j = c, i.e.
// dW row 128 ti + 4 i + ta (wide A; i for a narrow A), column 128 tj + 64 half + 2 j + tb (wide B; j for a narrow B).
// Extras: quad X+0 = odd column (dW[.][nb-1]) for the lane's TA rows, X+1 = bias for the same rows (both: half 0 only),
// X+2 = odd row (dW[na-1][.]) for the lane's TB columns, X+3 = {corner, corner bias, -, -} (half 0 only).
template <int TA, int TB>
__device__ __forceinline__ void t3_emit(const TnPair& pr, const T3Task& tk, int half, int Q, float4 s, int c, int kh, int lane,
                                        bool bad) {
    const bool f_xcol = tk.flags & TNF_XCOL, f_bias = tk.flags & TNF_BIAS, f_xrow = tk.flags & TNF_XROW;
    const int na_main = pr.na - (f_xrow ? 1 : 0), nb_main = pr.nb - (f_xcol ? 1 : 0);
    const float nan_ = __builtin_nanf("");
    const float e4[4] = {bad ? nan_ : s.x, bad ? nan_ : s.y, bad ? nan_ : s.z, bad ? nan_ : s.w};
    constexpr int NT = TA * TB * 4;
    if (Q < NT) {
        const int t = Q >> 2, g = Q & 3, ta = t / TB, tb = t % TB;
        const int j = TB == 2 ? 128 * tk.tj + 64 * half + 2 * c + tb : c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ii = e + 8 * g + 4 * kh;
            const int i = TA == 4 ? 128 * tk.ti + 4 * ii + ta : ii;
            if (i < na_main && j < nb_main) pr.G[(size_t)(pr.gn0 + i) * pr.ldg + pr.gk0 + j] = e4[e];
        }
        return;
    }
    if (kh != 0) return;          // the two k halves were summed into both; the lower half emits
    const int x = Q - NT;
    if (x == 0 || x == 1) {
        if (half != 0) return;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const int i = TA == 4 ? 128 * tk.ti + 4 * c + t : c;
            if (i < na_main) {
                if (x == 0 && f_xcol) pr.G[(size_t)(pr.gn0 + i) * pr.ldg + pr.gk0 + pr.nb - 1] = e4[t];
                if (x == 1 && f_bias) pr.bias_out[i] = e4[t];
            }
        }
    } else if (x == 2) {
        if (f_xrow) {
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int j = TB == 2 ? 128 * tk.tj + 64 * half + 2 * c + t : c;
                if (j < nb_main) pr.G[(size_t)(pr.gn0 + pr.na - 1) * pr.ldg + pr.gk0 + j] = e4[t];
            }
        }
    } else if (lane == 0 && half == 0 && f_xrow) {
        if (f_xcol) pr.G[(size_t)(pr.gn0 + pr.na - 1) * pr.ldg + pr.gk0 + pr.nb - 1] = e4[0];
        if (f_bias) pr.bias_out[pr.na - 1] = e4[1];
    }
}

// RUNS: the launch has a chunk-major operand -> steps are dealt to the row groups in runs of four and refilled per run (see below);
// otherwise step by step with immediate refills (a prefetch distance of the full ring: what small batches need)
template <int TA, int TB, bool RUNS>
__device__ __forceinline__ void t3_body(const T3Args& a, const T3Task& tk, int bx, float4* lds) {
    constexpr int NQ = t3_quads(TA, TB);
    constexpr int NH = TB == 2 ? 2 : 1;                 // column halves of the block
    constexpr int NR = T3_WAVES / NH;                   // row groups
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = NH == 2 ? (wave & 1) : 0, wr = NH == 2 ? (wave >> 1) : wave;
    const int c = lane & 31, kh = lane >> 5;
    const TnPair& pr = a.pair[tk.pair];
    const int M = a.M;
    // the block's row range: nsplit contiguous ranges whose length is a multiple of 8 x 8 rows (the last one is ragged)
    constexpr int GR = T3_ROWS * T3_WAVES;
    const int64_t per = ((int64_t)(M + tk.nsplit - 1) / tk.nsplit + GR - 1) / GR * GR;
    const int R0 = (int)min((int64_t)M, per * bx), R1 = (int)min((int64_t)M, per * (bx + 1));
    // columns of this lane, clamped into the row so every read is legal; columns past na / nb produce outputs that are
    // never stored
    const int acol = TA == 4 ? min(128 * tk.ti + 4 * c, pr.lda - 4) : min(c, pr.lda - 1);
    const int bcol = TB == 2 ? min(128 * tk.tj + 64 * half + 2 * c, pr.ldb - 2) : min(c, pr.ldb - 1);
    // B may be CHUNK-MAJOR ([ldb / 4 planes][rows][float4], what the big-graph hop kernel writes): element (row, col) then sits
    // at float offset ((col >> 2) * rows + row) * 4 + (col & 3) -- a row step is 4 floats, a lane's two adjacent columns still
    // share one float4
    const int rowB = pr.b_cm_rows > 0 ? 4 : pr.ldb;
    auto colB = [&](int col) { return pr.b_cm_rows > 0 ? (col >> 2) * pr.b_cm_rows * 4 + (col & 3) : col; };
    const int rowA = pr.a_cm_rows > 0 ? 4 : pr.lda;
    auto colA = [&](int col) { return pr.a_cm_rows > 0 ? (col >> 2) * pr.a_cm_rows * 4 + (col & 3) : col; };
    const float* Ap = pr.A + colA(acol);
    const float* Bp = pr.B + colB(bcol);
    const float* Xc = pr.B + colB(pr.nb - 1);     // the odd column of B (TNF_XCOL)
    const float* Yr = pr.A + colA(pr.na - 1);     // the odd column of A = odd row of dW (TNF_XROW)
    const float* Rs = pr.bias_rowscale;

    f32x16 acc[TA][TB];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ta][tb][q] = 0.f;
    float xc[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f}, xr[4] = {0.f, 0.f, 0.f, 0.f}, cn = 0.f, cb = 0.f;
    // Two-level summation for the NARROW shapes (a 4-wide operand: few accumulators, and -- being cheap -- few row splits, so an
    // accumulator would be ONE fp32 FMA chain over tens of thousands of rows at 414 k nodes: 2e-5 of the largest entry of dW2 on
    // a high-degree grid): every T3_FLUSH ring turns (1,024 rows of the wave) the accumulators are added into a second set and
    // cleared.  The wide shape has no registers for that and gets ~3x the row splits anyway.
    constexpr bool TWO_LEVEL = TA * TB < 8;
    constexpr int OA = TWO_LEVEL ? TA : 1, OB = TWO_LEVEL ? TB : 1;
    f32x16 outer[OA][OB];
    float oxc[4] = {0.f, 0.f, 0.f, 0.f}, obs[4] = {0.f, 0.f, 0.f, 0.f}, oxr[4] = {0.f, 0.f, 0.f, 0.f}, ocn = 0.f, ocb = 0.f;
#pragma unroll
    for (int ta = 0; ta < OA; ++ta)
#pragma unroll
        for (int tb = 0; tb < OB; ++tb)
#pragma unroll
            for (int q = 0; q < 16; ++q) outer[ta][tb][q] = 0.f;
    auto flush = [&]() {
        if (!TWO_LEVEL) return;
#pragma unroll
        for (int ta = 0; ta < OA; ++ta)
#pragma unroll
            for (int tb = 0; tb < OB; ++tb)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    outer[ta][tb][q] += acc[ta][tb][q];
                    acc[ta][tb][q] = 0.f;
                }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oxc[e] += xc[e]; xc[e] = 0.f;
            obs[e] += bs[e]; bs[e] = 0.f;
            oxr[e] += xr[e]; xr[e] = 0.f;
        }
        ocn += cn; cn = 0.f;
        ocb += cb; cb = 0.f;
    };

    // ---- the streaming loop: a RING of T3_R steps (2 rows each) per wave, hand-managed like gemm_nt's A fragment.
    // Every vector-memory instruction is inline asm and every wait a hand-counted `s_waitcnt vmcnt(N)` tied to the
    // registers it protects: left to hipcc, a slot's first use gets vmcnt(0) (a full drain of the prefetch) and the
    // prefetch distance is ONE batch -- measured 45-52 TF, latency-bound.  Here step t's operands are requested T3_R steps
    // (8 x 8 MFMAs x 64 cycles x 2 waves per SIMD ~ 8 k cycles) ahead of their use and refilled IN PLACE right after it;
    // VMEM returns in issue order, so "at most 5 (T3_R - 1) younger loads outstanding" is exact for the slot about to be
    // consumed.  A step's five loads: the A and B fragments and three per-row scalars of the VALU extras (the odd column of
    // B, the odd column of A, the bias rowscale), each a wave-uniform base + a constant per-lane offset (lane half kh reads
    // row 2t + kh).  All five are always issued, from always-valid addresses (results a task does not own are never
    // emitted), so the wait counts never depend on the task.
    const float* Rs2 = Rs ? Rs : pr.A;                 // no rowscale: any valid address, the value is replaced by 1
    const int rs_stride = Rs ? 1 : pr.lda;
    const bool has_rs = Rs != nullptr;
    typedef typename T3Frag<TA>::type FA;
    typedef typename T3Frag<TB>::type FB;
    struct Slot { FA a; FB b; float xv, yv, rs; };
    auto compute = [&](const Slot& t) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_at(t.a, ta), frag_at(t.b, tb), acc[ta][tb], 0, 0, 0);
        const float rsv = has_rs ? t.rs : 1.f;
#pragma unroll
        for (int e = 0; e < TA; ++e) {
            xc[e] = fmaf(frag_at(t.a, e), t.xv, xc[e]);
            bs[e] = fmaf(frag_at(t.a, e), rsv, bs[e]);
        }
#pragma unroll
        for (int e = 0; e < TB; ++e) xr[e] = fmaf(t.yv, frag_at(t.b, e), xr[e]);
        cn = fmaf(t.yv, t.xv, cn);
        cb = fmaf(t.yv, rsv, cb);
    };
    // rows of the block's range in steps of 2, dealt to the row groups in RUNS OF FOUR steps (8 consecutive rows): row group wr
    // takes the runs wr, wr + NR, ...  A chunk-major operand keeps 8 rows of a float4 plane in one 128-byte line, and the four
    // loads that share it are issued back to back (the refills go out per run, below) so they meet in L1; dealt step by step
    // (round 2) the line was touched by four different waves at four different times -- with 8 waves x 32 planes in flight it
    // was evicted in between (gemm_tn 2.72 -> 3.09 ms at 6470rte x 64 once its B operands became chunk-major).
    // `count` = this wave's full steps (both rows inside the range); the < 4 steps of a ragged last run go to that run's owner.
    const int nstep = (R1 - R0) >> 1;
    const int nrun = nstep >> 2, tail_steps = nstep & 3;
    const int count = RUNS ? 4 * (nrun > wr ? (nrun - wr + NR - 1) / NR : 0) + ((nrun % NR) == wr ? tail_steps : 0)
                           : (nstep > wr ? (nstep - wr + NR - 1) / NR : 0);
    // the wave's q-th step -> step of the block's range
    auto gstep = [&](int q) { return RUNS ? 4 * (NR * (q >> 2) + wr) + (q & 3) : wr + NR * q; };
    if (count > 0) {
        const uint32_t voA = (uint32_t)(kh * rowA + colA(acol)) * 4u, voB = (uint32_t)(kh * rowB + colB(bcol)) * 4u;
        const uint32_t voX = (uint32_t)(kh * rowB + colB(pr.nb - 1)) * 4u, voY = (uint32_t)(kh * rowA + colA(pr.na - 1)) * 4u;
        const uint32_t voR = (uint32_t)(kh * rs_stride) * 4u;
        const char* baseA = reinterpret_cast<const char*>(pr.A);
        const char* baseB = reinterpret_cast<const char*>(pr.B);
        const char* baseR = reinterpret_cast<const char*>(Rs2);
        auto issue = [&](Slot& t, int step) {   // refill IN PLACE; a step past the wave's last one re-reads the last one
            const int64_t row = R0 + 2 * (int64_t)gstep(min(step, count - 1));
            const char* pa = baseA + row * rowA * 4;
            const char* pb = baseB + row * rowB * 4;
            const char* prs = baseR + row * rs_stride * 4;
            if (TA == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(t.a) : "v"(voA), "s"(pa) : "memory");
            else asm volatile("global_load_dword %0, %1, %2" : "+v"(t.a) : "v"(voA), "s"(pa) : "memory");
            if (TB == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(t.b) : "v"(voB), "s"(pb) : "memory");
            else asm volatile("global_load_dword %0, %1, %2" : "+v"(t.b) : "v"(voB), "s"(pb) : "memory");
            asm volatile("global_load_dword %0, %1, %2" : "+v"(t.xv) : "v"(voX), "s"(pb) : "memory");
            asm volatile("global_load_dword %0, %1, %2" : "+v"(t.yv) : "v"(voY), "s"(pa) : "memory");
            asm volatile("global_load_dword %0, %1, %2" : "+v"(t.rs) : "v"(voR), "s"(prs) : "memory");
        };
        Slot ring[T3_R];
#pragma unroll
        for (int r = 0; r < T3_R; ++r) {
            frag_zero(ring[r].a);
            frag_zero(ring[r].b);
            ring[r].xv = ring[r].yv = ring[r].rs = 0.f;
            issue(ring[r], r);
        }
        for (int t0 = 0; t0 < count; t0 += T3_R) {
#pragma unroll
            for (int r = 0; r < T3_R; ++r) {
                // the slot about to be consumed has landed: exactly the 5 loads of each of the T3_R - 1 younger slots may still fly
                // (refills go out per run of four slots: slot r has the 5 loads of each of the 3 - (r & 3) later slots of its run
                //  and of the whole other run younger than its own)
                asm volatile("s_waitcnt vmcnt(%5)"
                             : "+v"(ring[r].a), "+v"(ring[r].b), "+v"(ring[r].xv), "+v"(ring[r].yv), "+v"(ring[r].rs)
                             : "n"(RUNS ? 5 * (T3_R - 1 - (r & 3)) : 5 * (T3_R - 1)));
                if (t0 + r < count) compute(ring[r]);
                if (!RUNS) {
                    issue(ring[r], t0 + r + T3_R);
                } else if ((r & 3) == 3) {
#pragma unroll
                    for (int rr = r - 3; rr <= r; ++rr) issue(ring[rr], t0 + rr + T3_R);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (TWO_LEVEL && ((t0 / T3_R) & (T3_FLUSH - 1)) == T3_FLUSH - 1) flush();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-fetched slots: nothing of the ring is in flight past here
    }
    // ragged tail: an odd row count leaves ONE row (kh = 0 only) for the row group whose turn it is -- compiler-visible
    // loads, the upper lane half contributes zeros
    if (((R1 - R0) & 1) && wr == (RUNS ? nrun : nstep) % NR) {
        const int row = R1 - 1;
        Slot t;
        t.a = *reinterpret_cast<const FA*>(Ap + (size_t)row * rowA);
        t.b = *reinterpret_cast<const FB*>(Bp + (size_t)row * rowB);
        t.xv = Xc[(size_t)row * rowB];
        t.yv = Yr[(size_t)row * rowA];
        t.rs = Rs2[(size_t)row * rs_stride];
        if (kh != 0) {
            frag_zero(t.b);
            t.xv = t.yv = t.rs = 0.f;
            frag_zero(t.a);
        }
        // (with has_rs false the bias column is the constant 1: the upper half must not add its 1 * a -- a is zeroed above)
        compute(t);
    }
    if (TWO_LEVEL) {   // total = second level + what the first level holds
        flush();
#pragma unroll
        for (int ta = 0; ta < OA; ++ta)
#pragma unroll
            for (int tb = 0; tb < OB; ++tb) acc[ta][tb] = outer[ta][tb];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xc[e] = oxc[e]; bs[e] = obs[e]; xr[e] = oxr[e];
        }
        cn = ocn; cb = ocb;
    }
    // the two k halves of the VALU extras
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        xc[e] += __shfl_xor(xc[e], 32);
        bs[e] += __shfl_xor(bs[e], 32);
        xr[e] += __shfl_xor(xr[e], 32);
    }
    cn += __shfl_xor(cn, 32);
    cb += __shfl_xor(cb, 32);

    // ---- everything a lane holds, as NQ float4 (see t3_emit for the order)
    float4 v[NQ];
#pragma unroll
    for (int Q = 0; Q < NQ - 4; ++Q) {
        const f32x16& t = acc[(Q >> 2) / TB][(Q >> 2) % TB];
        v[Q] = make_float4(t[4 * (Q & 3)], t[4 * (Q & 3) + 1], t[4 * (Q & 3) + 2], t[4 * (Q & 3) + 3]);
    }
    v[NQ - 4] = make_float4(xc[0], xc[1], xc[2], xc[3]);
    v[NQ - 3] = make_float4(bs[0], bs[1], bs[2], bs[3]);
    v[NQ - 2] = make_float4(xr[0], xr[1], xr[2], xr[3]);
    v[NQ - 1] = make_float4(cn, cb, 0.f, 0.f);
    // ---- fixed tree over the row groups (per column half), e.g. NR = 4: (0 + 2) + (1 + 3)
#pragma unroll
    for (int stride = NR / 2; stride >= 1; stride >>= 1) {
        if (wr >= stride && wr < 2 * stride) {
#pragma unroll
            for (int Q = 0; Q < NQ; ++Q) lds[(((wr - stride) * NH + half) * NQ + Q) * 64 + lane] = v[Q];
        }
        __syncthreads();
        if (wr < stride) {
#pragma unroll
            for (int Q = 0; Q < NQ; ++Q) {
                const float4 o = lds[((wr * NH + half) * NQ + Q) * 64 + lane];
                v[Q].x += o.x; v[Q].y += o.y; v[Q].z += o.z; v[Q].w += o.w;
                // left alone the scheduler hoists all NQ LDS reads above the adds: NQ x 4 more live registers -> spills
                if ((Q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (wr != 0) return;
    // ---- one block: write the gradient; several: publish the partial for tn_combine_kernel
    if (tk.nsplit == 1) {
        const bool bad = t3_bad(a);
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) t3_emit<TA, TB>(pr, tk, half, Q, v[Q], c, kh, lane, bad);
        return;
    }
    float4* mine = a.partial + tk.part0 + ((size_t)bx * NH + half) * NQ * 64;
#pragma unroll
    for (int Q = 0; Q < NQ; ++Q) st4_wt(reinterpret_cast<float*>(mine + Q * 64 + lane), v[Q]);   // (write-through: ~9 MB of partials
                                                                                                  //  that the combine launch would wait for)
}

template <bool RUNS>
__global__ __launch_bounds__(T3_THREADS, 1) void gemm_tn_kernel(const T3Args a, const DweRide ride) {
    extern __shared__ __attribute__((aligned(16))) float4 t3_lds[];
    if ((int)blockIdx.x >= a.nblocks) {   // riders: the dWe partial reductions of the same backward pass (edge.hip), 17 blocks per layer
        const int r = blockIdx.x - a.nblocks, bx_per = (ride.fe * ride.h + 15) / 16;
        const int job = r / bx_per;
        if (job < ride.njobs) dwe_reduce_body<T3_THREADS / 16>(ride.jobs.job[job], r - job * bx_per, ride.fe, ride.ld, ride.h,
                                                               reinterpret_cast<float (*)[17]>(t3_lds), t3_bad(a));
        return;
    }
    // task and row split of this workgroup.  Groups own consecutive id ranges; inside a group the ids run
    //     [chunk of 8 splits][member][split in chunk]
    // so the workgroups of ONE row range of all members (the 4 pairs of a TAGConv share dY, dP / dQ of an EdgeAggregation
    // share X) get ids that agree mod 8 and are dispatched together: workgroups are dealt round-robin to the 8 XCDs, each
    // with its own L2, and the shared operand is then fetched from HBM once per XCD instead of once per member.
    int t = 0;
    for (;;) {
        const int gs = a.task[t].gsize, nb = a.task[t].nsplit * gs;
        if ((int)blockIdx.x < a.task[t].block0 + nb || t + gs >= a.ntasks) break;
        t += gs;
    }
    const int local = blockIdx.x - a.task[t].block0, gs0 = a.task[t].gsize, ns0 = a.task[t].nsplit;
    const int nfull = ns0 >> 3, tail = ns0 & 7;   // the last chunk holds the < 8 left-over splits, unpadded (ids stay dense:
                                                  // padding ids would pin more work on some XCDs than on others)
    int member, bx;
    if (local < nfull * 8 * gs0) {
        const int chunk = local / (8 * gs0), rem = local - chunk * 8 * gs0;
        member = rem >> 3;
        bx = 8 * chunk + (rem & 7);
    } else {
        const int l2 = local - nfull * 8 * gs0;
        member = l2 / tail;
        bx = 8 * nfull + (l2 - member * tail);
    }
    const T3Task tk = a.task[t + member];
    if (tk.wa && tk.wb) t3_body<4, 2, RUNS>(a, tk, bx, t3_lds);
    else if (tk.wa) t3_body<4, 1, RUNS>(a, tk, bx, t3_lds);
    else if (tk.wb) t3_body<1, 2, RUNS>(a, tk, bx, t3_lds);
    else t3_body<1, 1, RUNS>(a, tk, bx, t3_lds);
}

// Second stage (tasks split over several blocks): thread = one (half, quad, lane) of one task; sums the task's partials in
// block order -> deterministic -- and emits into the nn.Linear layout.
__global__ __launch_bounds__(256) void tn_combine_kernel(const T3Args a) {
    const T3Task tk = a.task[blockIdx.y];
    if (tk.nsplit == 1) return;
    const int ta = tk.wa ? 4 : 1, tb = tk.wb ? 2 : 1, nh = tk.wb ? 2 : 1, NQ = t3_quads(ta, tb);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nh * NQ * 64) return;
    const int half = idx / (NQ * 64), Q = (idx >> 6) % NQ, lane = idx & 63;
    const float4* src = a.partial + tk.part0 + idx;
    const size_t stride = (size_t)nh * NQ * 64;
    // eight partials are REQUESTED at once (clamped index: a slot past the end re-reads the last partial and is dropped), then
    // added in block order: the kernel is one dependent-latency chain, ~5 us when the loads went out two at a time
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < tk.nsplit; b += 8) {
        float4 pp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pp[u] = src[(size_t)min(b + u, tk.nsplit - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool k = b + u < tk.nsplit;
            s.x = k ? s.x + pp[u].x : s.x; s.y = k ? s.y + pp[u].y : s.y;
            s.z = k ? s.z + pp[u].z : s.z; s.w = k ? s.w + pp[u].w : s.w;
        }
    }
    const TnPair& pr = a.pair[tk.pair];
    const int c = lane & 31, kh = lane >> 5;
    const bool bad = t3_bad(a);
    if (tk.wa && tk.wb) t3_emit<4, 2>(pr, tk, half, Q, s, c, kh, lane, bad);
    else if (tk.wa) t3_emit<4, 1>(pr, tk, half, Q, s, c, kh, lane, bad);
    else if (tk.wb) t3_emit<1, 2>(pr, tk, half, Q, s, c, kh, lane, bad);
    else t3_emit<1, 1>(pr, tk, half, Q, s, c, kh, lane, bad);
}

constexpr size_t T3_MAX_PARTIAL_F4 = (size_t)400 * 2 * t3_quads(4, 2) * 64;   // ~one workgroup per CU, with slack   // float4 capacity of the partial buffer

size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs) {
    (void)M; (void)max_na; (void)max_nb; (void)max_pairs;
    return T3_MAX_PARTIAL_F4 * 4 + 256;
}

static int launch_weight_grads_uniform(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s, const DweRide* ride,
                                       const int* stamp, int stamp_want);
int launch_weight_grads(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s, const DweRide* ride,
                        const int* stamp, int stamp_want) {
    // A launch with a chunk-major operand anywhere runs the runs-of-four kernel form for ALL its pairs (gemm_tn_kernel<true>:
    // shorter prefetch distance, 0.59 against 0.68 of the MFMA peak on row-major operands at 6470rte x 64).  A list that mixes
    // the two kinds (the big-graph configuration: the TAGConv pairs are chunk-major, the EdgeAggregation pairs are not) goes out as
    // two launches, each kind on its own kernel form, each planned for the whole chip.
    int ncm = 0;
    for (int q = 0; q < npairs; ++q) ncm += (pairs[q].a_cm_rows > 0 || pairs[q].b_cm_rows > 0) ? 1 : 0;
    if (ncm == 0 || ncm == npairs || M == 0) return launch_weight_grads_uniform(pairs, npairs, M, ws, s, ride, stamp, stamp_want);
    std::vector<TnPair> rm, cm;
    for (int q = 0; q < npairs; ++q) ((pairs[q].a_cm_rows > 0 || pairs[q].b_cm_rows > 0) ? cm : rm).push_back(pairs[q]);
    PFN_TRY(launch_weight_grads_uniform(cm.data(), (int)cm.size(), M, ws, s, ride, stamp, stamp_want));
    return launch_weight_grads_uniform(rm.data(), (int)rm.size(), M, ws, s, nullptr, stamp, stamp_want);
}
static int launch_weight_grads_uniform(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s, const DweRide* ride,
                                       const int* stamp, int stamp_want) {
    if (npairs == 0 || M == 0) {
        // no rows: every gradient is an empty sum
        for (int p = 0; p < npairs; ++p) {
            const TnPair& pr = pairs[p];
            for (int i = 0; i < pr.na; ++i)
                PFN_CHECK_HIP(hipMemsetAsync(pr.G + (size_t)(pr.gn0 + i) * pr.ldg + pr.gk0, 0, (size_t)pr.nb * sizeof(float), s));
            if (pr.bias_out) PFN_CHECK_HIP(hipMemsetAsync(pr.bias_out, 0, (size_t)pr.na * sizeof(float), s));
        }
        return PFN_OK;
    }
    const size_t cap_f4 = ws.floats >= 256 ? (ws.floats - 256) / 4 : 0;
    static std::atomic<uint64_t> lds_raised{0}, lds_raised_runs{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_tn_kernel<false>), T3_LDS_BYTES, lds_raised));
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_tn_kernel<true>), T3_LDS_BYTES, lds_raised_runs));
    const int want_env = 0;
    const int ncu = device_cus();
    bool ride_done = false;
    int p = 0;
    while (p < npairs) {
        T3Args ta;
        memset(&ta, 0, sizeof(ta));
        ta.M = (int)M;
        ta.partial = reinterpret_cast<float4*>(ws.partial);
        ta.stamp = stamp;
        ta.stamp_want = stamp_want;
        int np_here = 0;
        double cost_total = 0.0, flops = 0.0, bytes = 0.0;
        double cost[T3_MAX_TASKS];
        while (p < npairs && np_here < T3_MAX_PAIRS) {
            const TnPair& pr = pairs[p];
            if (pr.lda % 4 || pr.ldb % 4 || pr.lda < 4 || pr.ldb < 4 || pr.na < 1 || pr.nb < 1 || pr.na > pr.lda || pr.nb > pr.ldb) {
                set_error("launch_weight_grads: operand rows must be padded to a multiple of 4 floats (lda %d, ldb %d, na %d, nb %d)",
                          pr.lda, pr.ldb, pr.na, pr.nb);
                return PFN_EINVAL;
            }
            const bool wa = pr.na > 32, wb = pr.nb > 32;
            const bool xrow = wa && pr.na % 128 == 1, xcol = wb && pr.nb % 128 == 1;
            const int QA = wa ? (pr.na - (xrow ? 1 : 0) + 127) / 128 : 1, QB = wb ? (pr.nb - (xcol ? 1 : 0) + 127) / 128 : 1;
            if (QA * QB > T3_MAX_TASKS) {
                set_error("launch_weight_grads: a %d x %d weight needs %d tile tasks (> %d)", pr.na, pr.nb, QA * QB, T3_MAX_TASKS);
                return PFN_EINVAL;
            }
            if (ta.ntasks + QA * QB > T3_MAX_TASKS) break;
            for (int qi = 0; qi < QA; ++qi)
                for (int qj = 0; qj < QB; ++qj) {
                    T3Task& t = ta.task[ta.ntasks];
                    t.pair = (short)np_here;
                    t.ti = (short)qi;
                    t.tj = (short)qj;
                    t.wa = wa;
                    t.wb = wb;
                    t.flags = (short)(((xcol && qj == QB - 1) ? TNF_XCOL : 0) | ((pr.bias_out && qj == 0) ? TNF_BIAS : 0) |
                                      ((xrow && qi == QA - 1) ? TNF_XROW : 0));
                    // per row pair and block: MFMAs (16 / 4 / 1) plus a fixed per-step part (5 loads + the VALU extras per wave).
                    // Measured per row (tools/ubench/run_gemm_tn_ts.sh, case118v2 x 128): wide 140 cycles, wide A x narrow B 40,
                    // narrow A x wide B 51-61 (its waves 4-7 run a third behind the waves 0-3) -- at 6 / 18 of a wide task's share the
                    // two tasks of that shape ended 10 us after everything else (gemm_tn 108 -> 99 us at config 2 with 9)
                    cost[ta.ntasks] = (!wa && wb) ? 9.0 : (wa ? 4.0 : 1.0) * (wb ? 4.0 : 1.0) + 2.0;
                    cost_total += cost[ta.ntasks];
                    ++ta.ntasks;
                }
            flops += 2.0 * (double)M * pr.na * pr.nb;
            bytes += 4.0 * (double)M * (pr.na + pr.nb);
            ta.pair[np_here++] = pr;
            ++p;
        }
        // groups: consecutive single-tile tasks of the same shape that share A or B with the first of them (at most 8)
        for (int t = 0; t < ta.ntasks;) {
            int gsz = 1;
            const TnPair& p0 = ta.pair[ta.task[t].pair];
            const bool single = t + 1 >= ta.ntasks || ta.task[t + 1].pair != ta.task[t].pair;
            while (single && t + gsz < ta.ntasks && gsz < 8) {
                const T3Task& nx = ta.task[t + gsz];
                const TnPair& pn = ta.pair[nx.pair];
                const bool nx_single = (t + gsz + 1 >= ta.ntasks || ta.task[t + gsz + 1].pair != nx.pair) && nx.pair != ta.task[t + gsz - 1].pair;
                if (!nx_single || nx.wa != ta.task[t].wa || nx.wb != ta.task[t].wb || (pn.A != p0.A && pn.B != p0.B)) break;
                ++gsz;
            }
            for (int k = 0; k < gsz; ++k) {
                ta.task[t + k].gsize = (short)gsz;
                ta.task[t + k].gidx = (short)k;
            }
            t += gsz;
        }
        // row splits: about one workgroup per CU in total (two waves per SIMD), shared out in proportion to the tasks'
        // cost per row (equal inside a group); at least 64 rows per row group; when the partials would not fit the buffer the
        // whole plan is scaled down (never a single task: one task left unsplit would take M rows on one CU)
        const int max_split = (int)std::max<int64_t>(1, M / (64 * T3_WAVES));
        int nblocks = 0;
        size_t part = 0;
        // ... and never more workgroups than the chip runs at once (one per CU): the shares are rounded per task, and a plan of
        // 260 workgroups on 256 CUs is a second round for four of them -- +45 % (seen when two narrow pairs left the list)
        const int limit = want_env > 0 ? std::max(want_env, 1) : ncu;
        for (double want = limit; ; want *= 0.97) {
            nblocks = 0;
            part = 0;
            for (int t = 0; t < ta.ntasks; ++t) {
                T3Task& tk = ta.task[t];
                if (tk.gidx == 0) {
                    int ns = (int)(want * cost[t] / cost_total + 0.5);
                    ns = std::max(1, std::min(ns, max_split));
                    // an empty trailing range would leave its partial unwritten: shrink until every range holds rows
                    while (ns > 1 && round_up((M + ns - 1) / ns, T3_ROWS * T3_WAVES) * (ns - 1) >= M) --ns;
                    tk.nsplit = ns;
                    tk.block0 = nblocks;
                    nblocks += ns * tk.gsize;
                } else {
                    tk.nsplit = ta.task[t - tk.gidx].nsplit;
                    tk.block0 = ta.task[t - tk.gidx].block0;
                }
                tk.part0 = (int)part;
                if (tk.nsplit > 1) part += (size_t)tk.nsplit * (tk.wb ? 2 : 1) * t3_quads(tk.wa ? 4 : 1, tk.wb ? 2 : 1) * 64;
            }
            if ((part <= cap_f4 && nblocks <= limit) || want < 2.0) break;
        }
        if (part > cap_f4) {
            set_error("launch_weight_grads: reduction workspace too small");
            return PFN_ENOSPACE;
        }
        ta.nblocks = nblocks;
        {
            ProfScope ps("gemm_tn", bytes, flops, s);
            DweRide rd;
            int extra = 0;
            if (ride && !ride_done && ride->njobs > 0) {   // (the first launch of the pass carries the riders)
                rd = *ride;
                extra = ride->njobs * ((ride->fe * ride->h + 15) / 16);
                ride_done = true;
            }
            bool any_cm = false;   // a chunk-major operand anywhere in the launch: runs of four steps per row group
            for (int q = 0; q < np_here; ++q) any_cm |= ta.pair[q].b_cm_rows > 0 || ta.pair[q].a_cm_rows > 0;
            if (any_cm) gemm_tn_kernel<true><<<nblocks + extra, T3_THREADS, T3_LDS_BYTES, s>>>(ta, rd);
            else gemm_tn_kernel<false><<<nblocks + extra, T3_THREADS, T3_LDS_BYTES, s>>>(ta, rd);
            PFN_CHECK_LAUNCH();
        }
        bool any_split = false;
        for (int t = 0; t < ta.ntasks; ++t) any_split |= ta.task[t].nsplit > 1;
        if (any_split) {
            ProfScope ps("tn_reduce", 0.0, 0.0, s);
            tn_combine_kernel<<<dim3((2 * t3_quads(4, 2) * 64 + 255) / 256, ta.ntasks), 256, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
    }
    return PFN_OK;
}

}  // namespace pfn

