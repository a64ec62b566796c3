// Tall-skinny fp32 MFMA GEMM  C = sum_t A_t * B_t (+ fused epilogue)  for the per-node dense contractions of the
// hot path, and the weight re-layout ("pack") that feeds it (gfx950).
//
// Carries what the reference runs as torch addmm/mm per EDGE (EdgeAggregation.edge_aggr, networks/MPN.py:17-21,:28)
// and per node (TAGConv.lins, mask_embd :491-495), restructured to per-NODE products.  M = nodes
// (1e4..1e6), K and N <= a few hundred; exact fp32 on v_mfma_f32_32x32x2_f32 (there is no TF32/xf32 on gfx950).
//
//  pack : once per forward every weight is copied into a zero-padded "LDS image", for both orientations W and W^T:
//         per 32-column quarter q, per group g of four k's, 32 columns x 4 k's  ->  image[q][g][col][k & 3]; then the
//         up to 4 trailing columns (H = 129 = 4*32 + 1) as rem[g][col][k & 3].  K is padded to a multiple of 8 with
//         zeros.  A lane's B operands for four consecutive MFMA steps are one aligned 16-byte LDS read, and a quarter's
//         k range is one contiguous run of whole KiBs -> copied to LDS by global_load_lds, no VGPR round trip.
//  gemm : WEIGHT-STATIONARY.  The weights of ALL terms of a launch (<= 4 x 129 x 129 floats, for a slice of 128 / 64 /
//         32 output columns) are loaded into LDS ONCE per block (<= 160 KiB), then the block's 8 waves run free:
//         NO barrier and no LDS traffic other than B reads in the steady state.  Every wave owns 32-row tiles of A
//         (and CT = 1 or 2 quarters of 32 output columns), streams its A fragment straight from global into registers
//         (17 float4 per lane and piece, refilled for the next piece right after each chunk is consumed = a prefetch
//         distance of one whole piece), and strides over the row tiles persistently.  Inside an 8-wide k chunk lane half
//         kh = lane>>5 supplies k = 8m + 4kh + i at MFMA step i, so one 16-byte A load and one 16-byte B read feed four
//         steps.  Up to 4 trailing output columns never get an MFMA tile: they are VALU dot products off the fragments
//         the wave already holds (B values are LDS broadcasts).  The 32x32 accumulator layout puts 32 consecutive columns
//         of a row in one register across lanes; a DPP quad transpose turns that into four consecutive columns per lane,
//         so the fused epilogue (bias, deg*b2, residual, ReLU / dropout+ReLU, gradient gate) stores 16 bytes per lane
//         straight from registers.  The stores of one wave overlap the MFMAs of the other wave on its SIMD.
//         A launch whose weights do not fit in LDS even for 32-column slices (hidden_dim >> 129) is split into several
//         launches that accumulate into C (raw partial sums; the epilogue runs in the last one).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT_THREADS = 512;                 // (the streaming kernel's DMA deal assumes 8 waves)
constexpr int NT_WAVES = NT_THREADS / 64;
constexpr int NCH = 17;                         // eight-wide k chunks per piece
constexpr int KP = 8 * NCH;                     // 136 k's per piece: H = 129 is ONE piece
constexpr int NT_MAX_PIECES = 16;
constexpr int NT_LDS_BYTES = 160 * 1024;

// ------------------------------------------------------------------------------------------------ pack
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    // the dropout stream advances once per forward, before any kernel of that forward reads it
    if (a.rng_advance && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.rng_advance[1] += 1;
    if (a.stamp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.stamp = a.stamp_value;
    if (a.mask_count > 0) {   // the rider: pred_mask.float() (networks/MPN.py:533), spread over every block of the launch
        const int64_t nthr = (int64_t)gridDim.x * gridDim.y * blockDim.x;
        for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < a.mask_count; i += nthr)
            a.maskf[i] = a.mask_dtype == 0 ? (float)static_cast<const int64_t*>(a.mask)[i] : static_cast<const float*>(a.mask)[i];
    }
    slot_ea_body(a.slot_ea, ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x,
                 (int64_t)gridDim.x * gridDim.y * blockDim.x);
    if ((int)blockIdx.y >= a.njobs) return;
    pack_job_body(a.job[blockIdx.y], blockIdx.x, gridDim.x);
}

size_t packed_floats(int K, int ld_out) { return packed_fp32_floats(K, ld_out); }

int launch_pack(const PackJob* jobs, int njobs, uint64_t* rng_advance, hipStream_t s, const void* mask, int mask_dtype,
                float* maskf, int64_t mask_count, const SlotEa* slot_ea, int* stamp, int stamp_value) {
    if (mask && mask_dtype != 0 && mask_dtype != 1) {
        set_error("pred_mask dtype code %d unsupported (0: int64, 1: float32)", mask_dtype);
        return PFN_EINVAL;
    }
    for (int j0 = 0; j0 < njobs || (j0 == 0 && (mask || slot_ea || stamp)); j0 += PACK_MAX_JOBS) {
        PackArgs a;
        if (j0 == 0 && slot_ea) a.slot_ea = *slot_ea;
        a.njobs = std::max(0, std::min(PACK_MAX_JOBS, njobs - j0));
        a.rng_advance = j0 == 0 ? rng_advance : nullptr;
        a.stamp = j0 == 0 ? stamp : nullptr;
        a.stamp_value = stamp_value;
        const bool rider = j0 == 0 && mask != nullptr && mask_count > 0;
        a.mask = rider ? mask : nullptr;
        a.maskf = maskf;
        a.mask_count = rider ? mask_count : 0;
        a.mask_dtype = mask_dtype;
        long biggest = 0;
        for (int j = 0; j < a.njobs; ++j) {
            a.job[j] = jobs[j0 + j];
            biggest = std::max<long>(biggest, (long)packed_floats(jobs[j0 + j].K, jobs[j0 + j].ld_out));
        }
        const int bx = (int)std::min<long>(std::max<long>(1, (biggest + 255) / 256), 64);
        ProfScope ps("pack_weights", 0.0, 0.0, s);
        if (a.njobs == 0 && !rider && !a.slot_ea.ea_in && !a.stamp) continue;
        pack_weights_kernel<<<dim3(bx, std::max(1, a.njobs)), 256, 0, s>>>(a);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

// ------------------------------------------------------------------------------------------------- NT
// One 1 KiB LDS-DMA (64 lanes x 16 bytes; LDS destination = wave-uniform base + lane * 16).
// Inline asm on purpose: while hipcc can see an LDS-DMA in flight it waits vmcnt(0) -- not a counted vmcnt -- for every
// ordinary load it later needs.  Hidden from the compiler, the DMA is waited for by hand (dma_wait) before the barrier
// that publishes the weights; the compiler's own counted waits stay correct because VMEM returns in issue order.
__device__ __forceinline__ void dma_1k(const char* g, float* lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)lds_dst));
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(m0v)
        : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 4 x 4 transpose across the four lanes of a quad (DPP quad_perm, no LDS): on entry lane j of the quad holds v[i] =
// M[i][j], on exit v[i] = M[j][i].  Turns the 32x32 accumulator layout (one column per lane) into four consecutive
// columns of ONE row per lane, so the epilogue stores 16 bytes per lane (whole 128-byte lines per 8 lanes) instead of
// sixteen 4-byte stores -- dword stores are issue-bound and took ~3 us per 64x132 tile.
__device__ __forceinline__ float dpp_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose(float (&v)[4], int lane) {
    // (every result a plain select of two values: written as `if (o1) { v[0] = ..; v[2] = ..; } else { v[1] = ..; v[3] = ..; }` hipcc
    //  if-converts the assignments into "v[o1 ? 0 : 1] = .." -- a dynamic index, four compares and four selects per assignment;
    //  found in the ISA of the interleaved flush, round 6: ~12 vector instructions per register group, in every flush)
    const bool o1 = lane & 1, o2 = lane & 2;
    // exchange across lane ^ 1: (v0,v1) and (v2,v3)
    const float s01 = dpp_xor1(o1 ? v[0] : v[1]), s23 = dpp_xor1(o1 ? v[2] : v[3]);
    const float a0 = o1 ? s01 : v[0], a1 = o1 ? v[1] : s01, a2 = o1 ? s23 : v[2], a3 = o1 ? v[3] : s23;
    // exchange across lane ^ 2: (v0,v2) and (v1,v3)
    const float s02 = dpp_xor2(o2 ? a0 : a2), s13 = dpp_xor2(o2 ? a1 : a3);
    v[0] = o2 ? s02 : a0;
    v[1] = o2 ? s13 : a1;
    v[2] = o2 ? a2 : s02;
    v[3] = o2 ? a3 : s13;
}

// Per-element epilogue.  Every operand arrives BY VALUE (kernel-uniform scalars live in SGPRs; `aux` = rowscale[row] when
// the GEMM has a row-scaled bias, else gate/resid[row][col]; `cb`/`crb` = bias[col] / rowbias[col], zero when absent):
// reaching them through a pointer to the kernel-argument struct made hipcc emit per-lane waterfall loops -- ~2000
// instructions per flush.  All loads are done by the caller in one batch BEFORE its first store.
struct EpiCfg {
    int ncols, act;
    bool has_rowscale, has_resid, has_gate;
    float p_drop, keep_scale, gate_scale;
    DropKey dk;
};
struct NtPiece {
    // -- the eight dwords the stationary kernel's round loop reads of the NEXT piece, contiguous: one scalar load per round (they were
    //    fetched field by field, the current, the next and the refill piece each indexed at run time: three dependent scalar round
    //    trips between two multiplies, ~1 k cycles of the ~2 k a round spends outside nt_multiply -- with both waves of a SIMD in step
    //    at small M that is matrix-pipe idle time)
    const float* A;      // operand rows, advanced to this piece's first k
    int lda;             // floats between consecutive rows of A: its row stride -- or 4 when the operand is CHUNK-MAJOR
                         // ([ld / 4 planes][rows][float4]: what the big-graph hop kernel writes, edge.hip)
    int kscale;          // bytes between consecutive k's of one row: 4 -- or 4 * rows for a chunk-major operand (k a multiple of 4)
    int kmax;            // last legal 16-byte read position inside a row, relative to A
    int klen;            // k's of the piece: a multiple of 8, <= KP (the image is zero beyond the real K)
    int gl;              // group | 256 when the piece is the LAST of its group inside the launch (the flush follows it)
    int lds_off;         // float offset of the piece's slice image in LDS
    // -- prologue only
    const float* Bq;     // image of quarter 0 at this piece's first k group; quarter q lies q * qstride floats further
    const float* Brem;   // trailing-column image at this piece's first k group
    int qstride;         // floats between quarters of the image
    int group;           // output group
};
struct NtArgs {
    int M, ncols, ldc, npiece;
    int tps, cshift, nq, remv;   // quarters per LDS slice; log2(column groups per block); MFMA quarters; VALU columns (padded)
    int nrem, bias_group, ldr, ldg;
    int kuni, bias_lds_off;      // k length shared by every piece of the launch, or 0; float offset of the bias image in LDS
    int c_cm_rows, aux_cm_rows;  // > 0: C (every group) / the gate-or-residual tensor is CHUNK-major with that many rows per plane
    int reverse;                 // 1: the row tiles are walked from the last to the first (serpentine sweeps, pfn_internal.hpp)
    int klast, row0;             // 1: in every piece only the first MFMA step of the last chunk has real k's (K = 129); else 4.
                                 // row0: global index of row 0 (dropout counter)
    NtPiece piece[NT_MAX_PIECES];
    float* C[8];
    int gflags[8];               // per group: 1 = add the C already in memory, 2 = store raw sums (no epilogue).  int, not
                                 // char: a byte field of the kernel argument is fetched by a VECTOR load + vmcnt(0)
    const float *bias, *rowscale, *rowbias, *resid, *gate;
    const uint64_t* rng;
    uint32_t rng_stream;
    int act;
    float p_drop, gate_scale;
};

// ---- hand-managed VMEM.  In the steady state EVERY vector-memory instruction of a wave is inline asm, invisible to
// hipcc's waitcnt insertion, and every wait is a hand-counted `s_waitcnt vmcnt(N)` tied to the registers it protects:
//   * the A refill writes IN PLACE ("+v"): one 68-register fragment instead of the two sets the register allocator
//     keeps for a visible load (a spill anywhere in the flush costs a vmcnt(0) drain per reload -- measured 18 us/flush);
//   * a compiler-inserted wait would be vmcnt(0) (it cannot see the 17 younger prefetch loads) and drain the prefetch.
// VMEM returns in issue order and vmcnt counts loads and stores alike, so "wait until at most N younger ops are
// outstanding" is exact when N counts the ops issued after the one needed, and merely early when N is smaller.
__device__ __forceinline__ void vload_x4(f32x4& dst, const char* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ f32x4 vload_x4_addr(const float* p) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float vload_x1_addr(const float* p) {
    float r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float vload_x1_sv(const char* sbase, uint32_t voff) {
    float r;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
// wave-uniform 64-bit base (SGPRs) + 32-bit per-lane byte offset: the address costs the VECTOR unit nothing (see vstore_x4 for WT and the s_nop)
template <bool WT = false>
__device__ __forceinline__ void vstore_x4_sv(const char* sbase, uint32_t voff, f32x4 v) {
    if (WT) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// the same store under a lane mask held in SGPRs -- exec narrowed and restored INSIDE the asm block: no branch, so the store stays
// in the basic block of the MFMAs it is interleaved with (gemm_nt_kernel's ILF); a lane whose mask bit is clear stores nothing
__device__ __forceinline__ void vstore_x4_sv_masked(const char* sbase, uint32_t voff, f32x4 v, uint64_t mask) {
    uint64_t keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %4\n\tglobal_store_dwordx4 %1, %2, %3\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                 : "=&s"(keep)
                 : "v"(voff), "v"(v), "s"(sbase), "s"(mask)
                 : "memory");
}
template <bool WT = false>
__device__ __forceinline__ void vstore_x4(float* p, f32x4 v) {
    // the s_nop is the ISA's "VMEM store wider than 64 bits -> VALU overwrites its data registers" hazard (2 wait states),
    // which hipcc fills in for its own stores but cannot see inside inline asm (without it: intermittently wrong elements).
    // WT = WRITE-THROUGH (sc1), the small-M kernels (CT < 2: one row tile per wave, the latency regime): a kernel's plain stores
    // leave its output dirty in the XCD's L2 and the kernel boundary then waits for the write-back (MI355X_MICROARCH.md
    // "boundary": + B / 6 TB/s behind B dirty bytes) -- ~1 us per launch of a chain whose every link is 10-30 us long; written
    // through, the lines drain while the waves still multiply (back to back at 15,104 rows: 12.7 -> 11.9 / 18.9 -> 17.9 /
    // 29.5 -> 28.7 us for 1 / 2 / 4 terms; `nt`: no change).  At large M it buys nothing and costs a few per cent: plain.
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// float offset of element (row, col) of an activation tensor: row-major [rows][ld] -- or chunk-major [ld / 4 planes][cm_rows][float4]
// when cm_rows > 0 (col a multiple of 4 here: the kernels move float4s)
__device__ __forceinline__ size_t act_off(int row, int col, int ld, int cm_rows) {
    return cm_rows > 0 ? ((size_t)(col >> 2) * cm_rows + row) * 4 : (size_t)row * ld + col;
}
template <int N>
__device__ __forceinline__ void wait_a(f32x4& v) {   // the fragment chunk about to be consumed has landed
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// Multiply one LDS-resident piece into the accumulators and refill the A fragment for the next piece.
//   NFAST: 17 / 16 = the piece has exactly that many chunks -> straight-line code; 0 = generic (per-chunk guard).
//   LS   : MFMA steps of the LAST chunk that carry real k's (H = 129: k = 128 alone -> 1 step instead of 4; the other
//          three would multiply all-zero weight rows).
//   NR   : trailing VALU columns this wave accumulates (0, 1 or 4).
// Chunk m waits for a_cur[m] with vmcnt(16): after the load that filled it, the wave issued the 16 other refills of that
// round (plus, around a flush, a few stores / epilogue operands -- then the wait asks for slightly younger prefetches
// than strictly needed, never for the stores).
//   XW   : vector-memory ops the wave issues between the end of one piece's refills and the first chunk of the next (the
//          streaming kernel's 9 weight DMAs): they are younger than every pending fragment chunk, so every wait count grows by XW.
//   SL   : a functor called once per chunk, behind the chunk's MFMAs and in their basic block: the INTERLEAVED FLUSH (round 6,
//          gemm_nt_kernel's ILF) hands in the previous tile's epilogue, one register group per call, so that its vector
//          instructions issue between the wave's own MFMAs.
struct NtNoSlice {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int CT, int NR, int NFAST, int LS, int XW = 0, bool DIET = true, class SL = NtNoSlice>
__device__ __forceinline__ void nt_multiply(f32x16 (&acc)[CT > 0 ? CT : 1], float (&racc)[4], f32x4 (&a_cur)[NCH],
                                            const float* S, int klen, int tps, int tsel, uint32_t kh4, int r32,
                                            const char* nbase, uint32_t nvoff, int nkmax, uint32_t nkscale, SL&& slice = SL()) {
    constexpr bool FAST = NFAST != 0;
    constexpr int CTE = CT > 0 ? CT : 1, NRE = NR > 0 ? NR : 1;
    // the trailing column's weights are read one chunk ahead like the tiles' -- except under the interleaved flush, which has no
    // registers for the second set (12 short: hipcc then parks three chunks of the NEXT fragment in scratch): there they are read
    // at the head of their own chunk; only the trailing column's fma chain waits for them, never an MFMA
    constexpr bool SLICED = !std::is_same<typename std::decay<SL>::type, NtNoSlice>::value;
    constexpr bool R1BUF = false && SLICED;
    const int tile_floats = FAST ? NFAST * 8 * 32 : klen * 32;
    const float* Bt = S + tsel * tile_floats + (kh4 * 8 + r32) * 4;
    const float* Rt = S + tps * tile_floats + kh4 * 4;
    // B operands are software-pipelined ONE chunk ahead by hand, and a scheduling barrier closes every chunk: left to
    // itself the scheduler hoists the LDS reads of all 17 chunks to the top and spills.
    const uint32_t vlane0 = nvoff + nkscale * kh4;   // the lane's byte offset of chunk 0 of the NEXT piece's fragment (see the refills)
    f32x4 b_nxt[CTE], r_nxt[NRE];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b_nxt[ct] = *reinterpret_cast<const f32x4*>(Bt + ct * tile_floats);
    if (!R1BUF) {
#pragma unroll
        for (int c = 0; c < NR; ++c) r_nxt[c] = *reinterpret_cast<const f32x4*>(Rt + c * 4);
    }
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const bool live = FAST ? m < NFAST : 8 * m < klen;
        if (live) {
            f32x4 b[CTE], r[NRE];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) b[ct] = b_nxt[ct];
#pragma unroll
            for (int c = 0; c < NR; ++c) r[c] = R1BUF ? *reinterpret_cast<const f32x4*>(Rt + m * 32 + c * 4) : r_nxt[c];
            if (m + 1 < NCH && (FAST ? m + 1 < NFAST : 8 * (m + 1) < klen)) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b_nxt[ct] = *reinterpret_cast<const f32x4*>(Bt + ct * tile_floats + (m + 1) * 256);
                if (!R1BUF) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) r_nxt[c] = *reinterpret_cast<const f32x4*>(Rt + (m + 1) * 32 + c * 4);
                }
            }
            // (refills are issued per GROUP of four chunks, below: chunk m has 16 - (m & 3) younger loads)
            switch (m & 3) {
                case 0: wait_a<NCH - 1 + XW>(a_cur[m]); break;
                case 1: wait_a<NCH - 2 + XW>(a_cur[m]); break;
                case 2: wait_a<NCH - 3 + XW>(a_cur[m]); break;
                default: wait_a<NCH - 4 + XW>(a_cur[m]); break;
            }
            const f32x4 av = a_cur[m];
#pragma unroll
            for (int i = 0; i < ((FAST && m == NFAST - 1) ? LS : 4); ++i) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b[ct][i], acc[ct], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NR; ++c) racc[c] = fmaf(av[i], r[c][i], racc[c]);
            }
            slice(m);
            // Under the interleaved flush the trailing column's sum is consumed in a LATER basic block only (the tile is parked behind
            // the multiply), and LLVM's MachineSink then moves the WHOLE 65-fma chain there -- with all 17 weight vectors and a second
            // copy of the fragment (the refills overwrite the first) kept alive through the multiply: 136 + 68 registers, spills of
            // in-flight fragment registers.  Naming the sum here pins every chunk's fmas to their chunk.
            if (SLICED) {
#pragma unroll
                for (int c = 0; c < NR; ++c) asm volatile("" : "+v"(racc[c]));
            }
        }
        // chunks consumed: refill them IN PLACE for the next piece -- four at a time.  A 128-byte line of a row holds four
        // consecutive chunks (32 bytes each: the two lane halves); refilled one by one they are requested ~600 cycles apart,
        // and with 8 waves x 32 rows = 32 KB of distinct lines in flight per CU the line is evicted from L1 in between:
        // every chunk load went back to L2 (4x the L2 -> L1 traffic).  Issued back to back the four requests meet in L1.
        // Unconditional (the wait counts rely on exactly 17 refills per piece), from a clamped always-valid address: k's past
        // the row multiply zero B rows.  Address = wave-uniform 64-bit base (SGPRs) + 32-bit per-lane offset.
        if ((m & 3) == 3 || m == NCH - 1) {
#pragma unroll
            for (int mm = m & ~3; mm <= m; ++mm) {
                // VALU diet (round 4: an fp32 MFMA stream and the VALU do not overlap on a SIMD -- every vector instruction between
                // MFMAs is matrix-pipe time, profiles/r04_gemm_nt_cycle_accounting.txt): when only the LAST chunk of a piece can
                // reach past the operand's row (DIET: the launcher / caller guarantees it -- true for K <= 136; NOT for the last
                // piece of a piece-padded wide image, whose real k's end several chunks early), every other chunk takes the lane's
                // chunk-0 offset (one multiply-add per piece) and its distance from chunk 0 goes into the wave-uniform BASE (scalar
                // unit) -- was a min, a multiply and an add per refill: 51 vector instructions per piece and wave.  (Decided per
                // chunk at run time instead -- a uniform test of nkmax -- both forms of every refill exist and the fragment's
                // registers get copies: 90-140 spills.)
                if (FAST && DIET && mm < NFAST - 1) {
                    vload_x4(a_cur[mm], nbase + (size_t)mm * 8u * nkscale, vlane0);
                } else {
                    const uint32_t kk = min(kh4 + 8u * mm, (uint32_t)nkmax);
                    vload_x4(a_cur[mm], nbase, nvoff + nkscale * kk);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// VAR = the multiply variant of the launch (all pieces of a launch share it; chosen on the host, so that a kernel holds at most
// TWO instantiations of the round loop -- with / without the trailing VALU column, a per-wave property; with more of them in one
// kernel the CT = 2 build had no registers left for the one-step tail):
//   0: 17 chunks, the last chunk is ONE MFMA step (K = 129)     1: 17 chunks, last chunk 4 steps
//   2: 16 chunks (K = 128), no trailing column                  3: generic (per-chunk guards, 4 trailing columns; CT < 2 only)
//   PAIR: every piece's partial sum is formed on its own and the pieces are added in piece order (round 6), instead of ONE fp32 chain
//         through all k's of all terms.  A TAGConv product is a 516-deep sum; as one chain its rounding error is 2-4 x that of
//         "K + 1 products, added" -- which is what PyG's TAGConv, the oracle and gemm_nt_tiny_kernel compute (profiles/
//         r06_accumulation_order.txt: the whole distance between the HIP forward and the fp32 dataflow against float64).  Costs a
//         second accumulator set (16 registers per quarter): the CT = 1 kernels have them, the CT = 2 / streaming kernels (256
//         registers) do not.
//   ILF : INTERLEAVED FLUSH (round 6), for launches whose EVERY piece ends an output tile (S W2^T, P | Q, dS: one K = 129 piece per
//         32 x 64 tile) without per-element epilogue operands (no gate / residual / dropout / raw sums).  Measured: with the flush
//         compiled away these launches run 27-31 % faster (tools/ubench/run_gemm_nt_noflush.sh: 0.61 -> 0.83 of the fp32 MFMA peak in
//         the config-3 step) -- not because a flush is much work (~300 vector instructions and 8 stores per tile) but because the
//         SIMD partner's fp32 MFMA stream lets 0.6 vector and 0.12 vector-memory instructions per MFMA through to a wave that is
//         not itself multiplying (profiles/r04_gemm_nt_cycle_accounting.txt; tools/ubench/mfma_pace.hip: pausing the stream with
//         s_nop does not free the slots either -- they are blocked while the MFMA EXECUTES).  A wave's vector instructions do issue
//         between its OWN MFMAs.  So a finished tile is parked in a second accumulator set and flushed by the NEXT piece's multiply:
//         one register group (quad transpose, bias / row scale as one fma, ReLU as a max against 0 or -inf, one exec-masked store
//         inside an asm block: straight-line, no branch) behind the MFMAs of every other chunk, the trailing column behind chunk
//         15.  The fragment waits keep their counts: the slices' stores only make them stricter (vmcnt counts stores; a masked-off
//         store may not count, so no count may rely on one).  Same expressions, same order: bit-identical to the plain flush.
template <int CT, int VAR, bool PAIR = false, bool ILF = false>
__global__ __launch_bounds__(NT_THREADS, 1) void gemm_nt_kernel(const NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CTE = CT > 0 ? CT : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, kh = lane >> 5;
    // The column slices of one row group share an XCD (round 6).  Workgroups go to the eight XCDs round-robin in launch order
    // (blockIdx.x fastest), so (bx, slice 0) and (bx, slice 1) of a dim3(gx, 2) grid landed on XCDs bx % 8 and (bx + gx) % 8: every
    // operand row was fetched into TWO L2s (config 2: 83 MB of fabric traffic per launch for 39 MB of operands and results).  The
    // launch-order id is re-read as [chunk of 8 row groups][slice][row group in chunk]: the slices of a row group are launched
    // eight apart -- same XCD, same moment -- and the second one's fragment gathers hit that L2.  (A tail of gx % 8 row groups
    // keeps the old order.)
    int bx = blockIdx.x, slice = blockIdx.y;
    {
        const int gx = gridDim.x, ns = gridDim.y, full = gx & ~7;
        const int lin = blockIdx.y * gx + blockIdx.x;
        if (ns > 1 && lin < full * ns) {
            const int chunk = lin / (8 * ns), r = lin - chunk * 8 * ns;
            slice = r >> 3;
            bx = chunk * 8 + (r & 7);
        } else if (ns > 1) {
            const int t = lin - full * ns, tail = gx - full;
            slice = t / tail;
            bx = full + (t - slice * tail);
        }
    }
    const int cg = wave & ((1 << a.cshift) - 1);       // column group of this wave inside the slice
    const int rsub = wave >> a.cshift;                  // its row tile inside the block's step
    const int rw = NT_WAVES >> a.cshift;                // row tiles per block step
    const int tile0 = slice * a.tps + cg * CT;          // first 32-column quarter of this wave
    const bool mfma_on = CT > 0 && tile0 < a.nq;
    const bool rem_on = a.remv > 0 && slice == (int)gridDim.y - 1 && cg == 0;
    const int rem_col = 32 * a.nq;
    const int nrt = (a.M + 31) >> 5;
    const int rt_step = gridDim.x * rw;
    // Row tiles are dealt ROW-SLOT-MAJOR: tile t of a round goes to (block t % gridDim.x, row slot t / gridDim.x).  Full rounds are
    // unaffected; the LAST, partial round then hands its tiles to row slot 0 of every block first, then slot 1, ... -- one extra
    // tile per SIMD before any SIMD gets two (a block's waves w and w + 4 share a SIMD, and the matrix pipe of a SIMD is what a
    // round costs).  Dealt block-major (block b took tiles 8 b .. 8 b + 7) the partial round filled ALL waves of the first blocks:
    // at 7,552 row tiles (case118v2 x 2048) 7.375 rounds cost 8 on 48 CUs while the others idled -- now 7.5.
    int rt = rsub * (int)gridDim.x + bx;

    // A fragment addresses: uniform base of the row tile (64-bit, SGPRs) + per-lane byte offset of the lane's row inside
    // the tile (rows past M are clamped to the last row; their results are never stored)
    auto prt = [&](int t) { return a.reverse ? nrt - 1 - t : t; };   // the row tile a loop index stands for (serpentine sweeps)
    auto a_base = [&](int rt2, int p2) -> const char* {
        return reinterpret_cast<const char*>(a.piece[p2].A + (size_t)prt(rt2) * 32 * a.piece[p2].lda);
    };
    auto a_voff = [&](int rt2, int p2) -> uint32_t {
        const int lrow = min(r32, a.M - 1 - prt(rt2) * 32);
        return (uint32_t)(lrow * a.piece[p2].lda) * 4u;
    };

    // ---- first A fragment (in flight while the weights are copied)
    f32x4 a_cur[NCH];
    {
        const int rt0 = rt < nrt ? rt : 0;
        const char* b0 = a_base(rt0, 0);
        const uint32_t v0 = a_voff(rt0, 0);
        const int kmax0 = a.piece[0].kmax;
        const uint32_t ks0 = (uint32_t)a.piece[0].kscale;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const uint32_t kk = min((uint32_t)(8 * m + 4 * kh), (uint32_t)kmax0);
            a_cur[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            vload_x4(a_cur[m], b0, v0 + ks0 * kk);
        }
    }
    // ---- weights of every piece -> LDS, once: 1 KiB DMA pieces dealt round-robin to the 8 waves
    {
        int dealt = 0;
        for (int p = 0; p < a.npiece; ++p) {
            const int klen = a.piece[p].klen, kib = klen >> 3;
            for (int j = 0; j < a.tps; ++j) {
                const int q = slice * a.tps + j;
                if (q < a.nq) {
                    const char* src = reinterpret_cast<const char*>(a.piece[p].Bq + (size_t)q * a.piece[p].qstride) + lane * 16;
                    float* dst = lds + a.piece[p].lds_off + j * klen * 32;
                    for (int c = ((wave - dealt) % NT_WAVES + NT_WAVES) % NT_WAVES; c < kib; c += NT_WAVES) dma_1k(src + (c << 10), dst + (c << 8));
                    dealt += kib;
                }
            }
            if (tid < klen)
                *reinterpret_cast<float4*>(lds + a.piece[p].lds_off + a.tps * klen * 32 + tid * 4) =
                    *reinterpret_cast<const float4*>(a.piece[p].Brem + tid * 4);
        }
    }
    // ---- per-column epilogue operand (the bias -- or, for a row-scaled bias, the rowbias; a GEMM has one or the other,
    // checked by the launcher): staged in LDS, zero past ncols, and read back at flush time.  Held in registers for the
    // whole kernel it cost 8-12 VGPRs, the difference between spilling in the flush and not.
    {
        const float* bsrc = a.rowscale ? a.rowbias : a.bias;
        for (int i = tid; i < a.ldc; i += NT_THREADS) lds[a.bias_lds_off + i] = (bsrc && i < a.ncols) ? bsrc[i] : 0.f;
    }
    dma_wait();
    __syncthreads();   // the only barrier: from here on the waves run free
    if (rt >= nrt || !(mfma_on || rem_on)) return;

    EpiCfg ep;
    ep.ncols = a.ncols;
    ep.act = a.act;
    ep.has_rowscale = a.rowscale != nullptr;
    ep.has_resid = a.resid != nullptr;
    ep.has_gate = a.gate != nullptr;
    ep.p_drop = a.p_drop;
    ep.gate_scale = a.gate_scale;
    ep.dk = DropKey{0u, 0u, 0u, 0u};
    ep.keep_scale = 1.f;
    if (a.act == ACT_DROPOUT_RELU) {
        ep.dk = drop_key(a.rng[0], a.rng[1], a.rng_stream);
        ep.keep_scale = 1.0f / (1.0f - a.p_drop);
    }

    f32x16 acc[CTE];
#pragma unroll
    for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
    float racc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 psum[PAIR ? CTE : 1];                         // PAIR: the finished pieces of the current output tile
    float prsum[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PAIR) {
#pragma unroll
        for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
            for (int q = 0; q < 16; ++q) psum[ct][q] = 0.f;
    }
    // ILF: the parked tile -- sums, the scale its bias image takes per register group (the row scale, or 1 / 0 for a plain bias /
    // none), the floor of its activation, where it goes and which lanes store
    f32x16 il_acc[ILF ? CTE : 1];
    float il_racc = 0.f, il_sc = 0.f, il_lo = 0.f;   // il_sc: lane (r32, .) holds the scale of the parked tile's row r32
    uint64_t il_ok[4] = {0ull, 0ull, 0ull, 0ull}, il_okr = 0ull;
    float* il_C = a.C[0];
    int il_rbase = 0;
    // (addresses of the slices without a branch on the layout -- a branch would split the basic block the slice shares with its
    //  MFMAs: element (row, 32 q + c) lies at row * il_rs_ + q * il_qs_ + the lane's part, row-major and chunk-major alike)
    const int il_rs_ = a.c_cm_rows > 0 ? 4 : a.ldc, il_qs_ = a.c_cm_rows > 0 ? 32 * a.c_cm_rows : 32;
    const int il_rq_ = (rem_col >> 2) * (a.c_cm_rows > 0 ? 4 * a.c_cm_rows : 4);
    const uint32_t il_vc = (uint32_t)((r32 & 3) + 4 * kh) * (uint32_t)il_rs_ * 4u + (uint32_t)(r32 >> 2) * (a.c_cm_rows > 0 ? (uint32_t)a.c_cm_rows * 16u : 16u);
    const uint32_t il_vr = (uint32_t)r32 * (uint32_t)il_rs_ * 4u;
    if constexpr (ILF) {
#pragma unroll
        for (int ct = 0; ct < CTE; ++ct) {
#pragma unroll
            for (int q = 0; q < 16; ++q) il_acc[ct][q] = 0.f;
        }
        // the activation as ONE max: against 0 (ReLU) or against a quiet NaN (none: max(x, NaN) = x, and a NaN x stays NaN -- the
        // poison of an unvalidated topology must come through a GEMM without activation as it does through the plain flush)
        il_lo = a.act == ACT_RELU ? 0.f : __builtin_nanf("");
    }
    const int nr = rem_on ? (a.nrem > 1 ? 4 : 1) : 0;   // trailing columns this wave owns (the generic path always computes 4)
    const float* extra = a.gate ? a.gate : a.resid;     // at most one of rowscale / gate / resid per GEMM
    const int ldx = a.gate ? a.ldg : a.ldr;
    uint32_t kh4 = 4u * kh;
    // The round loop is instantiated ONCE per multiply variant and the variant is chosen outside it (all pieces of a
    // launch normally share one k length): with several variants merging inside the loop, the fragment's loop-carried
    // registers are copied to working registers and back every round (a second 68-register set, spills in the flush).
    auto rounds = [&](auto nfast_c, auto nr_c, auto ls_c) {
    constexpr int NFAST = decltype(nfast_c)::value, NR = decltype(nr_c)::value, LS = decltype(ls_c)::value;
    int p = 0;
    // (the piece table read through the kernel-argument segment's own address: indexed as `a.piece[pi]` with all eight fields wanted
    //  at once, hipcc loads the WHOLE table into SGPRs up front and spills it -- 108-164 SGPR spills, a v_readlane per use)
    typedef const NtPiece __attribute__((address_space(4)))* NtPieceK;
    const NtPieceK pk = (NtPieceK)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(NtArgs, piece));
    int cur_gl = pk[0].gl, cur_lds = pk[0].lds_off, cur_klen = pk[0].klen;
    while (true) {
        asm volatile("" : "+v"(kh4));   // opaque per round: keeps the 17 refill offsets from being hoisted into 17 VGPRs
        // ---- the piece after this one: same row tile, next piece -- or the wave's next row tile, first piece
        int np = p + 1, nrt_ = rt;
        if (np == a.npiece) {
            np = 0;
            nrt_ = rt + rt_step;
        }
        const bool more = nrt_ < nrt;
        const int group = cur_gl & 255;
        const bool flush_after = !more || (cur_gl >> 8) != 0;
        const int pi = more ? np : p;                        // no next piece: refill from the current one (harmless)
        // the next piece's eight dwords: one load, nothing of it is needed before the refills inside the multiply (the current
        // piece's group / LDS offset / length came with the load of the round before)
        const NtPieceK nx = pk + pi;
        const float* nxA = nx->A;
        const int nx_lda = nx->lda, nx_kscale = nx->kscale, nkmax = nx->kmax, nx_klen = nx->klen, nx_gl = nx->gl, nx_lds = nx->lds_off;
        const int nrt2 = prt(more ? nrt_ : rt);
        const char* nbase = reinterpret_cast<const char*>(nxA + (size_t)nrt2 * 32 * nx_lda);
        const uint32_t nvoff = (uint32_t)(min(r32, a.M - 1 - nrt2 * 32) * nx_lda) * 4u;
        // ---- per-row epilogue operands of the flush that follows this piece: requested NOW (hidden loads), so they are
        // older than the 17 refills issued inside the multiply; the flush waits for them with vmcnt(17)
        const int rbase = prt(rt) * 32;
        // Addresses of the flush (these loads and the stores): lane (u = r32 >> 2, j = r32 & 3, kh) holds, after the quad transpose,
        // row rbase + j + 4 kh + 8 g and the four columns 32 q + 4 u .. of quarter q for register group g -- the lane's part of the
        // address does not depend on the tile, the group or the quarter, so it is ONE 32-bit offset per tensor and everything else
        // goes into a wave-uniform base (scalar unit).  Was: a 64-bit row x stride product, a clamp and an add per load / store on
        // the vector unit -- which an fp32 MFMA stream does not overlap with (profiles/r04_gemm_nt_cycle_accounting.txt).  Rows
        // past M (the last tile) and columns past the row (a partial last quarter) are masked out instead of clamped.
        const int jrow = (r32 & 3) + 4 * kh, uc4 = r32 & ~3;
        const bool tile_full = rbase + 32 <= a.M;
        auto rowok = [&](int g) { return rbase + jrow + 8 * g < a.M; };
        auto colok = [&](int q) { return 32 * q + uc4 < a.ldc; };
        auto qfull = [&](int q) { return 32 * q + 32 <= a.ldc; };
        // float offset of (row rbase + 8 g, column 32 q) of a tensor with row stride ld / chunk-major with cm rows per plane
        auto ubase = [&](int q, int g, int ld, int cm) -> size_t {
            return cm > 0 ? ((size_t)8 * q * cm + rbase + 8 * g) * 4 : (size_t)(rbase + 8 * g) * ld + 32 * q;
        };
        auto lane_off = [&](int ld, int cm) -> uint32_t {   // bytes
            return cm > 0 ? ((uint32_t)(uc4 >> 2) * (uint32_t)cm + jrow) * 16u : ((uint32_t)jrow * (uint32_t)ld + uc4) * 4u;
        };
        // (cleared as whole vectors of independent components, every round: defined by an empty asm instead -- no instruction -- the
        //  component-wise loads below no longer coalesce into them and the copy into place runs before the hidden load has landed;
        //  cleared only under has_aux, one register of the CT = 2 kernels spills)
        f32x4 aux[CTE][4], raux;
        float il_rs = 0.f;   // ILF: the tile's row scales, row r32's in lane (r32, .) (requested here, used by the NEXT multiply: a register
                             // group's four rows are fetched from their lanes by the slice -- 2 registers instead of 10)
        if constexpr (ILF) {
            if (a.rowscale)   // (clamped rows; loaded IN PLACE: a fresh output register would be merged into il_rs by a copy that runs
                              //  before the hidden load has landed)
                asm volatile("global_load_dword %0, %1, %2" : "+v"(il_rs)
                             : "v"((uint32_t)min(r32, a.M - 1 - rbase) * 4u), "s"(reinterpret_cast<const char*>(a.rowscale + rbase)) : "memory");
        }
#pragma unroll
        for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) aux[ct][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        raux = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has_aux = !ILF && flush_after && (a.rowscale || extra);
        if (has_aux) {
            if (a.rowscale) {
                // (one dword per row, loaded straight into its component of `aux` -- unconditionally, from a clamped row: a masked load
                //  would go through a temporary, and the copy out of it would run before the hidden load has landed)
                const char* rb = reinterpret_cast<const char*>(a.rowscale + rbase);
                const int rlast = a.M - 1 - rbase;
#pragma unroll
                for (int g = 0; g < 4; ++g) aux[0][g][0] = vload_x1_sv(rb, (uint32_t)min(jrow + 8 * g, rlast) * 4u);
                if (rem_on) raux[0] = vload_x1_sv(rb, (uint32_t)min(r32, rlast) * 4u);
            } else {
                const uint32_t vx = lane_off(ldx, a.aux_cm_rows);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int q = tile0 + ct < a.nq ? tile0 + ct : 0;   // (a quarter past the last one: loads quarter 0, never stored)
                    const bool cfull = qfull(q);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if ((tile_full || rowok(g)) && (cfull || colok(q)))
                            vload_x4(aux[ct][g], reinterpret_cast<const char*>(extra + ubase(q, g, ldx, a.aux_cm_rows)), vx);
                }
                if (rem_on && (tile_full || rbase + r32 < a.M))
                    vload_x4(raux, reinterpret_cast<const char*>(extra + (a.aux_cm_rows > 0 ? ((size_t)(rem_col >> 2) * a.aux_cm_rows + rbase) * 4
                                                                                         : (size_t)rbase * ldx + rem_col)),
                             a.aux_cm_rows > 0 ? (uint32_t)r32 * 16u : (uint32_t)r32 * (uint32_t)ldx * 4u);
            }
        }
        // ---- multiply
        {
            const float* S = lds + cur_lds;
            const int klen = cur_klen, tsel = cg * CT;
            if constexpr (ILF) {
                // one register group of the parked tile per call: (quarter u >> 2, group u & 3) behind chunk 2 u, the trailing
                // column behind chunk 15 (chunk 16 is two MFMAs long)
                auto il_unit = [&](int u) {
                    const int ct = u >> 2, g = u & 3;
                    float t[4] = {il_acc[ct][4 * g], il_acc[ct][4 * g + 1], il_acc[ct][4 * g + 2], il_acc[ct][4 * g + 3]};
                    quad_transpose(t, lane);
                    const int q = tile0 + ct;
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(lds + a.bias_lds_off + 32 * q + (r32 & ~3));   // (LDS: 8 registers less)
                    const float sc = __shfl(il_sc, jrow + 8 * g);   // the scale of this lane's row of the group
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = fmaxf(fmaf(sc, cb[e], t[e]), il_lo);
                    const size_t ub = (size_t)(il_rbase + 8 * g) * il_rs_ + (size_t)q * il_qs_;
                    vstore_x4_sv_masked(reinterpret_cast<const char*>(il_C + ub), il_vc, f32x4{t[0], t[1], t[2], t[3]}, il_ok[g]);
                };
                auto il_rem = [&]() {
                    const float tot = il_racc + __shfl_xor(il_racc, 32);   // the two k halves
                    const float x = fmaxf(fmaf(il_sc, lds[a.bias_lds_off + rem_col], tot), il_lo);
                    const size_t ub = (size_t)il_rbase * il_rs_ + il_rq_;
                    vstore_x4_sv_masked(reinterpret_cast<const char*>(il_C + ub), il_vr, f32x4{x, 0.f, 0.f, 0.f}, il_okr);
                };
                auto slice = [&](int m) {
                    if (m < 4 * CT * 2 && (m & 1) == 0) il_unit(m >> 1);
                    if (NR > 0 && m == 15) il_rem();
                };
                nt_multiply<CT, NR, NFAST, LS, 0, true>(acc, racc, a_cur, S, klen, a.tps, tsel, kh4, r32, nbase, nvoff, nkmax,
                                                        (uint32_t)nx_kscale, slice);
                // ---- the tile just multiplied is parked (every piece of an ILF launch ends one); the next multiply flushes it
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[CTE - 1]));   // (XDL write -> VALU read, see the flush)
                const bool use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
                if (a.rowscale) {   // the row scales requested before the multiply: the 17 refills (and the slices' stores) are younger
                    asm volatile("s_waitcnt vmcnt(17)" : "+v"(il_rs));
                    il_sc = il_rs;
                } else {
                    il_sc = use_bias ? 1.f : 0.f;
                }
#pragma unroll
                for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        il_acc[ct][q] = acc[ct][q];
                        acc[ct][q] = 0.f;
                    }
                il_racc = racc[0];
                racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
                il_C = a.C[group];
                il_rbase = rbase;
#pragma unroll
                for (int g = 0; g < 4; ++g) il_ok[g] = __ballot(rbase + jrow + 8 * g < a.M);
                il_okr = __ballot(rem_on && kh == 0 && rbase + r32 < a.M);
                if (!more) {   // the wave's last tile: flushed here
#pragma unroll
                    for (int u = 0; u < 4 * CT; ++u) {
                        il_unit(u);
                        __builtin_amdgcn_sched_barrier(0);   // (one unit's temporaries at a time)
                    }
                    if (NR > 0) il_rem();
                    break;
                }
                p = np;
                rt = nrt_;
                cur_gl = nx_gl;
                cur_lds = nx_lds;
                cur_klen = nx_klen;
                continue;
            }
            nt_multiply<CT, NR, NFAST, LS>(acc, racc, a_cur, S, klen, a.tps, tsel, kh4, r32, nbase, nvoff, nkmax,
                                           (uint32_t)nx_kscale);
        }
        if (flush_after) {
            // ---- flush straight from registers: acc[q] of lane (r32, kh) is D[row (q&3) + 8 (q>>2) + 4 kh][col r32];
            // after the quad transpose lane (u = r32 >> 2, j = r32 & 3) holds, for register group g, row
            // rbase + j + 8 g + 4 kh and the four columns col0 .. col0 + 3
            // The last MFMA was issued a few instructions ago and its 16 passes are still writing the accumulators; hipcc's
            // hazard recognizer does not look past the inline asm that closes the multiply, so the >= 18 wait states an
            // XDL write needs before a VALU read are spent by hand (without them: intermittently stale accumulator rows).
            if (CT > 0) {
                if (CT == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[CTE - 1]));
                else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]));
            }
            if constexpr (PAIR) {   // (earlier pieces) + this one: the pieces in piece order
#pragma unroll
                for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        acc[ct][q] = psum[ct][q] + acc[ct][q];
                        psum[ct][q] = 0.f;
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    racc[e] = prsum[e] + racc[e];
                    prsum[e] = 0.f;
                }
            }
            float* C = a.C[group];
            const int gf = a.gflags[group];
            const bool use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
            if (has_aux) {   // the operands requested before the multiply: exactly the 17 refills are younger
                if (CT == 2)
                    asm volatile("s_waitcnt vmcnt(17)" : "+v"(aux[0][0]), "+v"(aux[0][1]), "+v"(aux[0][2]), "+v"(aux[0][3]),
                                 "+v"(aux[CTE - 1][0]), "+v"(aux[CTE - 1][1]), "+v"(aux[CTE - 1][2]), "+v"(aux[CTE - 1][3]), "+v"(raux));
                else
                    asm volatile("s_waitcnt vmcnt(17)" : "+v"(aux[0][0]), "+v"(aux[0][1]), "+v"(aux[0][2]), "+v"(aux[0][3]), "+v"(raux));
                if (a.rowscale) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float rs = aux[0][g][0];
#pragma unroll
                        for (int ct = 0; ct < CTE; ++ct) aux[ct][g] = f32x4{rs, rs, rs, rs};
                    }
                    raux = f32x4{raux[0], raux[0], raux[0], raux[0]};
                }
            }
            // Every epilogue stage is ONE kernel-uniform branch around straight-line code for all 16 values of a tile (a
            // per-element `switch` cost 7 us per flush).  Columns past ncols need no masking: packed weights, bias and
            // rowbias are zero there and the pad columns of resid / gate are zero by the layout invariant, so every stage
            // maps 0 to 0.
            const bool raw = gf & 2;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col0 = 32 * (tile0 + ct) + (r32 & ~3);
                if (tile0 + ct < a.nq && col0 < a.ldc) {
                    float v[4][4];
                    // row / destination of register group g are recomputed where they are used (one multiply-add each):
                    // kept in arrays they were 12 more registers alive through every epilogue stage -- with CT = 2 the
                    // difference between spilling and not, and a spill reload anywhere in the kernel makes hipcc put an
                    // s_waitcnt vmcnt(0) -- a drain of the 17 prefetched A chunks -- at the head of EVERY piece
                    const int row_base = rbase + (r32 & 3) + 4 * kh;
                    auto row_of = [&](int g) { return row_base + 8 * g; };
                    auto dst_of = [&](int g) {
                        const int rw_ = row_of(g);
                        return C + act_off(rw_ < a.M ? rw_ : a.M - 1, col0, a.ldc, a.c_cm_rows);
                    };
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4] = {acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
                        quad_transpose(t, lane);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = t[e];
                    }
                    if (gf & 1) {   // accumulating launch (weights larger than LDS): rare, visible loads
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 old = *reinterpret_cast<const float4*>(dst_of(g));
                            v[g][0] += old.x; v[g][1] += old.y; v[g][2] += old.z; v[g][3] += old.w;
                        }
                    }
                    if (!raw) {
                        const f32x4 cb4 = *reinterpret_cast<const f32x4*>(lds + a.bias_lds_off + col0);
                        if (use_bias && !ep.has_rowscale) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] += cb4[e];
                        }
                        if (ep.has_rowscale) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] = fmaf(aux[ct][g][e], cb4[e], v[g][e]);
                        }
                        if (ep.has_resid) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] += aux[ct][g][e];
                        }
                        if (ep.act == ACT_RELU) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                        } else if (ep.act == ACT_DROPOUT_RELU) {
                            // (opaque: the column group is loop-invariant per lane, and hipcc otherwise hoists the Philox
                            //  rounds' invariant parts out of the tile loop -- per-lane values alive through the whole
                            //  kernel, i.e. spills)
                            uint32_t cgv = (uint32_t)(col0 >> 2);
                            asm volatile("" : "+v"(cgv));
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float u[4];
                                dropout_uniform4(ep.dk, (uint32_t)(row_of(g) + a.row0), cgv, u);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] = (u[e] >= ep.p_drop && v[g][e] > 0.f) ? v[g][e] * ep.keep_scale : 0.f;
                            }
                        }
                        if (ep.has_gate) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[g][e] = aux[ct][g][e] > 0.f ? v[g][e] * ep.gate_scale : 0.f;
                        }
                    }
                    {   // uniform base + the lane's one offset (see the epilogue-operand loads); a full tile of a full quarter needs no mask
                        const int q = tile0 + ct;
                        const uint32_t vc = lane_off(a.ldc, a.c_cm_rows);
                        const bool unmasked = tile_full && qfull(q);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const char* cb = reinterpret_cast<const char*>(C + ubase(q, g, a.ldc, a.c_cm_rows));
                            if (unmasked || (rowok(g) && colok(q))) vstore_x4_sv<(CT < 2)>(cb, vc, f32x4{v[g][0], v[g][1], v[g][2], v[g][3]});
                        }
                    }
                }
            }
            if (rem_on) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = racc[e] + __shfl_xor(racc[e], 32);   // the two k halves
                const int row = rbase + r32;
                if (kh == 0 && row < a.M) {
                    float* dst = C + act_off(row, rem_col, a.ldc, a.c_cm_rows);
                    if (gf & 1) {
                        const float4 old = *reinterpret_cast<const float4*>(dst);
                        v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
                    }
                    if (!raw) {
                        const f32x4 rcb = *reinterpret_cast<const f32x4*>(lds + a.bias_lds_off + rem_col);
                        float ud[4] = {1.f, 1.f, 1.f, 1.f};
                        if (ep.act == ACT_DROPOUT_RELU) {
                            uint32_t cgv = (uint32_t)(rem_col >> 2);
                            asm volatile("" : "+v"(cgv));
                            dropout_uniform4(ep.dk, (uint32_t)(row + a.row0), cgv, ud);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = v[e] + ((use_bias && !ep.has_rowscale) ? rcb[e] : 0.f);
                            if (ep.has_rowscale) x = fmaf(raux[e], rcb[e], x);
                            if (ep.has_resid) x += raux[e];
                            if (ep.act == ACT_RELU) {
                                x = fmaxf(x, 0.f);
                            } else if (ep.act == ACT_DROPOUT_RELU) {
                                x = (ud[e] >= ep.p_drop && x > 0.f) ? x * ep.keep_scale : 0.f;
                            }
                            if (ep.has_gate) x = raux[e] > 0.f ? x * ep.gate_scale : 0.f;
                            v[e] = rem_col + e < ep.ncols ? x : 0.f;
                        }
                    }
                    vstore_x4<(CT < 2)>(dst, f32x4{v[0], v[1], v[2], v[3]});
                }
            }
#pragma unroll
            for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
            racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
        }
        if constexpr (PAIR) {
            if (!flush_after) {   // a piece of the tile is done, more follow: its sum joins the finished ones, the chain starts anew
                if (CT > 0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]));   // (XDL write -> VALU read, see the flush)
#pragma unroll
                for (int ct = 0; ct < CTE; ++ct)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        psum[ct][q] += acc[ct][q];
                        acc[ct][q] = 0.f;
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    prsum[e] += racc[e];
                    racc[e] = 0.f;
                }
            }
        }
        if (!more) break;
        p = np;
        rt = nrt_;
        cur_gl = nx_gl;
        cur_lds = nx_lds;
        cur_klen = nx_klen;
    }
    };
    using std::integral_constant;
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    using I4 = integral_constant<int, 4>;
    if constexpr (VAR == 0) {
        if (nr == 0) rounds(integral_constant<int, NCH>{}, I0{}, I1{});
        else rounds(integral_constant<int, NCH>{}, I1{}, I1{});
    } else if constexpr (VAR == 1) {
        if (nr == 0) rounds(integral_constant<int, NCH>{}, I0{}, I4{});
        else rounds(integral_constant<int, NCH>{}, I1{}, I4{});
    } else if constexpr (VAR == 2) {
        rounds(integral_constant<int, NCH - 1>{}, I0{}, I4{});
    } else {
        rounds(I0{}, I4{}, I4{});
    }
    // CT == 2 exists only for the straight-line variants (the launcher never pairs it with the generic one: two tiles plus
    // four trailing columns plus per-chunk guards do not fit the register file without spills)
}


// ------------------------------------------------------------------------------------- NT, weight-STREAMING (large M)
// The weight-stationary kernel above reads the A stream once per LDS slice: the 4 x 129 x 129 weights of a TAGConv product do not
// fit LDS whole (287 KB), so its 64-column slices read every operand row TWICE -- 8.2 bytes per cycle and CU at the MFMA rate,
// against ~10 the memory system delivers: at 414 k rows the kernel sat at 0.60 of the MFMA peak, 17 % of that the A stream.  When
// M is large every wave has many row tiles, and the trade flips: here a wave owns FULL rows (all four 32-column quarters and the
// trailing column: 64 accumulator registers) of one 32-row tile per round, reads its A rows ONCE, and the WEIGHTS stream through
// LDS instead -- one 129 x 132 term image (72 KB) at a time, double-buffered: while the block multiplies piece s out of buffer
// s & 1, the DMA of piece s + 1 lands in the other one (the whole 287 KB weight set is L2-resident: every block streams the same
// bytes).  One barrier per piece: it publishes the DMA'd image and retires the buffer the previous piece was read from.  Per
// chunk a wave now issues 16 MFMAs per fragment wait, refill group and LDS address update instead of 8.
// vmcnt bookkeeping (VMEM returns in order): per piece a wave issues 9 DMAs (right behind the barrier) and 17 fragment refills
// (inside the multiply), so at chunk m the ops younger than the fragment chunk it needs are 16 - (m & 3) refills + 9 DMAs
// (nt_multiply's XW), and the DMAs of the NEXT piece are complete once at most the 17 refills issued after them are outstanding.
// Only for launches of the shape the large graphs produce: every piece K = 129 (one 136-k piece, one-step tail), 129 output
// columns (4 quarters + 1 trailing column).
constexpr int WS_IMG = KP * (32 * 4) + 768;                  // floats per LDS buffer: 4 quarter images + 3 KiB for the 2,176-byte
                                                             // trailing-column image (copied as three 1 KiB DMAs)
constexpr int WS_DMAS = 9;                                   // DMAs per wave and piece: 4 x 17 + 3 = 71 -> 72 slots over 8 waves
// REM / LS: the H = 129 shape (4 quarters + the trailing column, one real step in a piece's last chunk: <true, 1>), or a WIDE
// product (round 4: hidden_dim 512 = the reference's configs/large.json) -- no trailing column, output columns in slices of 128
// (blockIdx.y), every piece a full 136-k one thanks to the piece-padded images (k8_of): <false, 4>.  Such a product does not fit
// the stationary kernel's LDS even as 32-column slices (4 terms x 512 k); it used to run as several launches accumulating raw
// sums into C with every operand row read once per 64-column slice (0.40 of the MFMA peak at case118v2 x 128).
template <bool REM, int LS>
__global__ __launch_bounds__(NT_THREADS, 1) void gemm_nt_ws_kernel(const NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CT = 4;
    const int cq0 = 4 * (int)blockIdx.y, c0 = 128 * (int)blockIdx.y;   // first quarter / column of this block's slice
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, kh = lane >> 5;
    const int nrt = (a.M + 31) >> 5;
    const int rt_step = gridDim.x * NT_WAVES;
    const int nround = (nrt + rt_step - 1) / rt_step;       // every wave of every block runs the same number of pieces (barriers)
    const int total = nround * a.npiece;
    int rt = blockIdx.x * NT_WAVES + wave;
    const int rem_col = 128;
    const int bias_off = 2 * WS_IMG;

    auto prt = [&](int t) { return a.reverse ? nrt - 1 - t : t; };   // the row tile a loop index stands for (serpentine sweeps)
    auto a_base = [&](int rt2, int p2) -> const char* {
        const int rc = prt(rt2 < nrt ? rt2 : nrt - 1);      // a wave past the last row tile keeps pace on the last tile; it stores nothing
        return reinterpret_cast<const char*>(a.piece[p2].A + (size_t)rc * 32 * a.piece[p2].lda);
    };
    auto a_voff = [&](int rt2, int p2) -> uint32_t {
        const int rc = prt(rt2 < nrt ? rt2 : nrt - 1);
        const int lrow = min(r32, a.M - 1 - rc * 32);
        return (uint32_t)(lrow * a.piece[p2].lda) * 4u;
    };
    // the 72 one-KiB DMA slots of one piece image, dealt round-robin: slot = wave + 8 j; slots 0..67 = quarter (slot / 17), KiB
    // (slot % 17); 68..70 = the trailing-column image's three KiB (it is 2,176 bytes: the over-read stays inside the packed-weight
    // buffer / workspace and lands in the 3 KiB the LDS buffer reserves); slot 71 repeats slot 68 so that every wave issues 9
    auto issue_dma = [&](int p2, int buf) {
        float* dst0 = lds + buf * WS_IMG;
        const char* bq = reinterpret_cast<const char*>(a.piece[p2].Bq) + lane * 16;
        const char* br = reinterpret_cast<const char*>(a.piece[p2].Brem) + lane * 16;
        const size_t qbytes = (size_t)a.piece[p2].qstride * 4;
#pragma unroll
        for (int j = 0; j < WS_DMAS; ++j) {
            const int slot = wave + NT_WAVES * j;
            if (slot < 68 || !REM) {                        // (no trailing column: slots 68..71 repeat slots 0..3 -- every wave issues 9)
                const int s2 = slot < 68 ? slot : slot - 68;
                const int q = s2 / 17, c = s2 - 17 * q;
                dma_1k(bq + (cq0 + q) * qbytes + ((size_t)c << 10), dst0 + q * (KP * 32) + (c << 8));
            } else {
                const int c = slot == 71 ? 0 : slot - 68;
                dma_1k(br + ((size_t)c << 10), dst0 + 4 * (KP * 32) + (c << 8));
            }
        }
    };

    // ---- first A fragment, first weight image, bias image
    f32x4 a_cur[NCH];
    {
        const char* b0 = a_base(rt, 0);
        const uint32_t v0 = a_voff(rt, 0);
        const int kmax0 = a.piece[0].kmax;
        const uint32_t ks0 = (uint32_t)a.piece[0].kscale;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const uint32_t kk = min((uint32_t)(8 * m + 4 * kh), (uint32_t)kmax0);
            a_cur[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            vload_x4(a_cur[m], b0, v0 + ks0 * kk);
        }
    }
    issue_dma(0, 0);
    {
        const float* bsrc = a.rowscale ? a.rowbias : a.bias;
        for (int i = tid; i < a.ldc; i += NT_THREADS) lds[bias_off + i] = (bsrc && i < a.ncols) ? bsrc[i] : 0.f;
    }

    EpiCfg ep;
    ep.ncols = a.ncols;
    ep.act = a.act;
    ep.has_rowscale = a.rowscale != nullptr;
    ep.has_resid = a.resid != nullptr;
    ep.has_gate = a.gate != nullptr;
    ep.p_drop = a.p_drop;
    ep.gate_scale = a.gate_scale;
    ep.dk = DropKey{0u, 0u, 0u, 0u};
    ep.keep_scale = 1.f;
    if (a.act == ACT_DROPOUT_RELU) {
        ep.dk = drop_key(a.rng[0], a.rng[1], a.rng_stream);
        ep.keep_scale = 1.0f / (1.0f - a.p_drop);
    }
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
    float racc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* extra = a.gate ? a.gate : a.resid;         // at most one of rowscale / gate / resid per GEMM
    const int ldx = a.gate ? a.ldg : a.ldr;
    uint32_t kh4 = 4u * kh;

    int p = 0;
    for (int s = 0; s < total; ++s) {
        asm volatile("" : "+v"(kh4));   // opaque per round: keeps the 17 refill offsets from being hoisted into 17 VGPRs
        int np = p + 1, nrt_ = rt;
        if (np == a.npiece) {
            np = 0;
            nrt_ = rt + rt_step;
        }
        const bool last = s + 1 == total;
        const int group = a.piece[p].group;
        const bool flush_after = last || np == 0 || a.piece[np].group != group;
        // ---- piece s has landed in buffer s & 1 (this wave's share: the first piece is the youngest thing in flight; later the
        // 17 refills of the previous multiply are younger), every wave is done reading the other buffer: one barrier says both
        if (s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        issue_dma(last ? p : np, (s + 1) & 1);              // (nothing follows the last piece: a harmless copy keeps the counts)
        // ---- multiply out of buffer s & 1, refilling the fragment for the next piece / row tile
        {
            const int pi = last ? p : np, rti = last ? rt : nrt_;
            nt_multiply<CT, REM ? 1 : 0, NCH, LS, WS_DMAS, REM>(acc, racc, a_cur, lds + (s & 1) * WS_IMG, KP, 4, 0, kh4, r32, a_base(rti, pi),
                                                           a_voff(rti, pi), a.piece[pi].kmax, (uint32_t)a.piece[pi].kscale);
        }
        if (flush_after) {
            // the accumulators are still being written by the last MFMAs (see the stationary kernel's flush)
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            const int rbase = prt(rt < nrt ? rt : nrt - 1) * 32;
            const bool live = rt < nrt;
            float* C = a.C[group];
            const bool use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
            const bool has_aux = a.rowscale || extra;
            // per-row epilogue operands, all requested at once (the B operand registers of the multiply are dead here), one wait
            f32x4 aux[CT][4], raux;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) aux[ct][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            raux = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_aux) {
                if (a.rowscale) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = rbase + (r32 & 3) + 8 * g + 4 * kh;
                        aux[0][g][0] = vload_x1_addr(a.rowscale + (row < a.M ? row : a.M - 1));
                    }
                    const int row = rbase + r32;
                    if (REM) raux[0] = vload_x1_addr(a.rowscale + (row < a.M ? row : a.M - 1));
                } else {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int col0 = c0 + 32 * ct + (r32 & ~3);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int row = rbase + (r32 & 3) + 8 * g + 4 * kh;
                            aux[ct][g] = vload_x4_addr(extra + act_off(row < a.M ? row : a.M - 1, col0, ldx, a.aux_cm_rows));
                        }
                    }
                    const int row = rbase + r32;
                    if (REM) raux = vload_x4_addr(extra + act_off(row < a.M ? row : a.M - 1, rem_col, ldx, a.aux_cm_rows));
                }
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(aux[0][0]), "+v"(aux[0][1]), "+v"(aux[0][2]), "+v"(aux[0][3]), "+v"(aux[1][0]), "+v"(aux[1][1]),
                               "+v"(aux[1][2]), "+v"(aux[1][3]), "+v"(aux[2][0]), "+v"(aux[2][1]), "+v"(aux[2][2]), "+v"(aux[2][3]),
                               "+v"(aux[3][0]), "+v"(aux[3][1]), "+v"(aux[3][2]), "+v"(aux[3][3]), "+v"(raux));
                if (a.rowscale) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float rs = aux[0][g][0];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) aux[ct][g] = f32x4{rs, rs, rs, rs};
                    }
                    raux = f32x4{raux[0], raux[0], raux[0], raux[0]};
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col0 = c0 + 32 * ct + (r32 & ~3);
                float v[4][4];
                const int row_base = rbase + (r32 & 3) + 4 * kh;
                auto row_of = [&](int g) { return row_base + 8 * g; };
                auto dst_of = [&](int g) {
                    const int rw_ = row_of(g);
                    return C + act_off(rw_ < a.M ? rw_ : a.M - 1, col0, a.ldc, a.c_cm_rows);
                };
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float t[4] = {acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
                    quad_transpose(t, lane);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = t[e];
                }
                const f32x4 cb4 = *reinterpret_cast<const f32x4*>(lds + bias_off + col0);
                if (use_bias && !ep.has_rowscale) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] += cb4[e];
                }
                if (ep.has_rowscale) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = fmaf(aux[ct][g][e], cb4[e], v[g][e]);
                }
                if (ep.has_resid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] += aux[ct][g][e];
                }
                if (ep.act == ACT_RELU) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                } else if (ep.act == ACT_DROPOUT_RELU) {
                    uint32_t cgv = (uint32_t)(col0 >> 2);
                    asm volatile("" : "+v"(cgv));
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float u[4];
                        dropout_uniform4(ep.dk, (uint32_t)(row_of(g) + a.row0), cgv, u);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = (u[e] >= ep.p_drop && v[g][e] > 0.f) ? v[g][e] * ep.keep_scale : 0.f;
                    }
                }
                if (ep.has_gate) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = aux[ct][g][e] > 0.f ? v[g][e] * ep.gate_scale : 0.f;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (live && row_of(g) < a.M) vstore_x4(dst_of(g), f32x4{v[g][0], v[g][1], v[g][2], v[g][3]});
            }
            if (REM) {   // the trailing column (129th): the two k halves, lane half 0 stores
                float v0 = racc[0] + __shfl_xor(racc[0], 32);
                const int row = rbase + r32;
                if (kh == 0 && live && row < a.M) {
                    const f32x4 rcb = *reinterpret_cast<const f32x4*>(lds + bias_off + rem_col);
                    float x = v0 + ((use_bias && !ep.has_rowscale) ? rcb[0] : 0.f);
                    if (ep.has_rowscale) x = fmaf(raux[0], rcb[0], x);
                    if (ep.has_resid) x += raux[0];
                    if (ep.act == ACT_RELU) {
                        x = fmaxf(x, 0.f);
                    } else if (ep.act == ACT_DROPOUT_RELU) {
                        float ud[4];
                        uint32_t cgv = (uint32_t)(rem_col >> 2);
                        asm volatile("" : "+v"(cgv));
                        dropout_uniform4(ep.dk, (uint32_t)(row + a.row0), cgv, ud);
                        x = (ud[0] >= ep.p_drop && x > 0.f) ? x * ep.keep_scale : 0.f;
                    }
                    if (ep.has_gate) x = raux[0] > 0.f ? x * ep.gate_scale : 0.f;
                    vstore_x4(C + act_off(row, rem_col, a.ldc, a.c_cm_rows), f32x4{x, 0.f, 0.f, 0.f});
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
            racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
        }
        p = np;
        rt = nrt_;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last DMA / refills: nothing of this block is in flight past here
}


// ------------------------------------------------------------------------------------- NT, small M (latency)
// A batch of ONE graph (118 rows: the per-sample latency the reference itself measures, perfomance_evaluator.py:61-74) is four row
// tiles: the kernels above give each wave a whole tile x all terms -- 260 dependent MFMAs = 7 us on one SIMD while 250 CUs idle --
// behind a 135 KB weight copy.  Here the K dimension is split ACROSS WAVES: block = one (row tile, 32-column quarter), wave p =
// term p: its A fragment and its quarter of the term's weight image straight from global memory into registers (34 float4 per lane,
// both requested at once), 65 MFMAs, the partial tiles summed through LDS in term order, wave 0 runs the epilogue.  The trailing
// column (129th) is a VALU dot product in the blocks of quarter 0.  Only the shapes that path produces: every piece K = 129, 129
// output columns, bias or row-scaled bias, optional ReLU, one output.
constexpr int TINY_MAX_PIECES = 8;
__global__ __launch_bounds__(64 * TINY_MAX_PIECES) void gemm_nt_tiny_kernel(const NtArgs a) {
    __shared__ __attribute__((aligned(16))) float part[TINY_MAX_PIECES][17][64];   // [term][16 accumulator values + trailing column][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, kh = lane >> 5;
    const int rt = blockIdx.x, q = blockIdx.y;
    const NtPiece& pc = a.piece[p];
    const int lrow = min(r32, a.M - 1 - rt * 32);
    const float* arow = pc.A + (size_t)(rt * 32) * pc.lda;
    f32x4 av[NCH], bv[NCH], rv[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const uint32_t kk = min((uint32_t)(8 * m + 4 * kh), (uint32_t)pc.kmax);
        av[m] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(arow) + (size_t)lrow * pc.lda * 4 + (size_t)pc.kscale * kk);
        bv[m] = *reinterpret_cast<const f32x4*>(pc.Bq + (size_t)q * pc.qstride + ((2 * m + kh) * 32 + r32) * 4);
    }
    const bool rem = q == 0;
    if (rem) {
#pragma unroll
        for (int m = 0; m < NCH; ++m) rv[m] = *reinterpret_cast<const f32x4*>(pc.Brem + ((2 * m + kh) * 4 + 0) * 4);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float racc = 0.f;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
#pragma unroll
        for (int i = 0; i < (m == NCH - 1 ? 1 : 4); ++i) {       // K = 129: the last chunk carries one real k
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][i], bv[m][i], acc, 0, 0, 0);
            if (rem) racc = fmaf(av[m][i], rv[m][i], racc);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[p][i][lane] = acc[i];
    part[p][16][lane] = racc;
    __syncthreads();
    if (p != 0) return;
    float v16[16];
    float rsum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = 0.f;
    for (int t2 = 0; t2 < a.npiece; ++t2) {                      // term order: deterministic
#pragma unroll
        for (int i = 0; i < 16; ++i) v16[i] += part[t2][i][lane];
        rsum += part[t2][16][lane];
    }
    // ---- epilogue (as gemm_nt_kernel's flush: quad transpose -> four consecutive columns of one row per lane)
    const int rbase = rt * 32;
    float* C = a.C[0];
    const float* cvec = a.rowscale ? a.rowbias : a.bias;
    const int col0 = 32 * q + (r32 & ~3);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float t4[4] = {v16[4 * g], v16[4 * g + 1], v16[4 * g + 2], v16[4 * g + 3]};
        quad_transpose(t4, lane);
        const int row = rbase + (r32 & 3) + 8 * g + 4 * kh;
        if (row < a.M) {
            const float rs = a.rowscale ? a.rowscale[row] : 1.f;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = col0 + e;
                float x = t4[e] + ((cvec && col < a.ncols) ? rs * cvec[col] : 0.f);
                if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                o[e] = col < a.ncols ? x : 0.f;
            }
            *reinterpret_cast<float4*>(C + act_off(row, col0, a.ldc, a.c_cm_rows)) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (rem) {
        const float tot = rsum + __shfl_xor(rsum, 32);           // the two k halves
        const int row = rbase + r32;
        if (kh == 0 && row < a.M) {
            const float rs = a.rowscale ? a.rowscale[row] : 1.f;
            float x = tot + (cvec ? rs * cvec[128] : 0.f);
            if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
            *reinterpret_cast<float4*>(C + act_off(row, 128, a.ldc, a.c_cm_rows)) = make_float4(x, 0.f, 0.f, 0.f);
        }
    }
}

template <int CT, int VAR, bool PAIR = false, bool ILF = false>
static int launch_variant(const NtArgs& k, dim3 grid, size_t lds_bytes, hipStream_t s) {
    static std::atomic<uint64_t> lds_raised{0};
    PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_nt_kernel<CT, VAR, PAIR, ILF>), NT_LDS_BYTES, lds_raised));
    gemm_nt_kernel<CT, VAR, PAIR, ILF><<<grid, NT_THREADS, lds_bytes, s>>>(k);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}
// the multiply variant of one launch (see gemm_nt_kernel); -1: not expressible with this CT
static int pick_variant(const NtArgs& k, int CT) {
    const int nr = k.remv > 0 ? (k.nrem > 1 ? 4 : 1) : 0;   // trailing columns the launch's last column group owns
    bool diet_ok = true;                                     // (see launch_gemm_nt_rows: the straight-line variants need it)
    for (int i = 0; i < k.npiece; ++i) diet_ok = diet_ok && k.piece[i].kmax >= k.piece[i].klen - 12;
    if (k.kuni == KP && nr <= 1 && diet_ok) return k.klast == 1 ? 0 : 1;
    if (k.kuni == KP - 8 && nr == 0 && diet_ok) return 2;
    return CT < 2 ? 3 : -1;
}

static int launch_gemm_nt_rows(const GemmArgs& a, hipStream_t s, bool top);
int launch_gemm_nt(const GemmArgs& a, hipStream_t s) { return launch_gemm_nt_rows(a, s, true); }
// top: the caller's whole product (profiled as ONE gemm_nt; may be cut into a streaming launch over whole rounds of row tiles and
// a stationary launch over the rest).  !top: the tail launch of such a cut -- stationary kernel, no profile bracket of its own.
static int launch_gemm_nt_rows(const GemmArgs& a, hipStream_t s, bool top) {
    if (a.M == 0) return PFN_OK;
    int remv, nq;
    col_plan(a.ldc, remv, nq);
    if (a.ldc % 4) {
        set_error("gemm_nt: output row stride %d must be a multiple of 4", a.ldc);
        return PFN_EINVAL;
    }
    if (a.bias && a.rowscale) {
        set_error("gemm_nt: a GEMM takes a bias or a row-scaled bias, not both");
        return PFN_EINVAL;
    }
    double flops = 0.0, bytes = (double)a.ngroup * a.M * a.ncols * 4.0;
    std::vector<NtPiece> pieces;
    std::vector<int> last_steps;   // per piece: MFMA steps of its last chunk that carry real k's (only meaningful for 136-k pieces)
    for (int t = 0; t < a.nterm; ++t) {
        const GemmTerm& tm = a.term[t];
        if (tm.lda % 4 != 0 || tm.lda < ((tm.K + 3) & ~3)) {
            set_error("gemm_nt: operand row stride %d must be a multiple of 4 and >= roundup(K=%d, 4)", tm.lda, tm.K);
            return PFN_EINVAL;
        }
        if (tm.Bp == nullptr) {
            set_error("gemm_nt: term %d has no packed weight", t);
            return PFN_EINVAL;
        }
        if (t > 0 && tm.group < a.term[t - 1].group) {
            set_error("gemm_nt: terms must be sorted by output group");
            return PFN_EINVAL;
        }
        flops += 2.0 * a.M * tm.K * a.ncols;
        bytes += (double)a.M * tm.K * 4.0;
        const int K8 = k8_of(tm.K), qstride = (K8 >> 2) * 128;
        for (int k0 = 0; k0 < K8; k0 += KP) {
            NtPiece pc;
            pc.A = tm.cm_rows > 0 ? tm.A + (size_t)(k0 >> 2) * tm.cm_rows * 4 : tm.A + k0;
            pc.Bq = tm.Bp + (size_t)(k0 >> 2) * 128;
            pc.Brem = tm.Bp + (size_t)nq * qstride + (size_t)(k0 >> 2) * 16;
            pc.lda = tm.cm_rows > 0 ? 4 : tm.lda;
            pc.kscale = tm.cm_rows > 0 ? tm.cm_rows * 4 : 4;
            pc.kmax = tm.lda - 4 - k0;
            pc.klen = std::min(KP, K8 - k0);
            pc.qstride = qstride;
            pc.group = tm.group;
            pc.gl = tm.group;
            pc.lds_off = 0;
            pieces.push_back(pc);
            {
                const int real_tail = std::min(tm.K - k0, KP) - (KP - 8);   // real k's in chunk 16 of a full piece
                last_steps.push_back(real_tail >= 1 && real_tail <= 4 ? real_tail : 4);
            }
        }
    }
    // ---- how many 32-column quarters of every piece fit in LDS at once (tps), and where the launch has to be cut
    auto piece_bytes = [](const NtPiece& pc, int tps) { return (size_t)pc.klen * (32 * tps + 4) * sizeof(float); };
    const size_t bias_bytes = (size_t)round_up((int64_t)a.ldc * sizeof(float), 16);   // the bias image rides behind the pieces
    const size_t lds_budget = (size_t)NT_LDS_BYTES - bias_bytes;
    int tps = 0;
    if (nq > 0) {
        const int tps_cap = 4;
        const int start = std::min(nq >= 3 ? 4 : nq, std::max(1, tps_cap));
        for (tps = start; tps >= 1; tps >>= 1) {
            size_t tot = 0;
            for (const NtPiece& pc : pieces) tot += piece_bytes(pc, tps);
            if (tot <= lds_budget && pieces.size() <= (size_t)NT_MAX_PIECES) break;
        }
        if (tps < 1) tps = std::min(start, 2);   // does not fit whole: several accumulating launches
    }
    const int ncu = device_cus();
    const int nrt = (a.M + 31) / 32;
    const int nslices = tps > 0 ? (nq + tps - 1) / tps : 1;
    // wave tile: two quarters per wave halve the A re-reads, but only when there is enough work to fill the chip twice over
    int CT = tps == 0 ? 0 : 1;
    if (pieces.empty()) {
        set_error("gemm_nt: no terms");
        return PFN_EINVAL;
    }
    bool fast = true;
    for (const NtPiece& pc : pieces) fast &= pc.klen == pieces[0].klen;
    const int nrem = std::max(0, std::min(remv, a.ncols - 32 * nq));
    fast = fast && ((pieces[0].klen == KP && nrem <= 1) || (pieces[0].klen == KP - 8 && remv == 0));
    // the straight-line variants refill every chunk but a piece's last without a clamp (nt_multiply's DIET): only the last chunk may
    // reach past the operand's row.  False for the later pieces of a piece-padded image (K > 136): those take the generic variant.
    bool diet_ok = true;
    for (const NtPiece& pc : pieces) diet_ok = diet_ok && pc.kmax >= pc.klen - 12;   // lane half 1 of chunk klen / 8 - 2 reads k = klen - 12 ..
    fast = fast && diet_ok;
    if (fast && tps >= 2 && (long)nrt * nslices * (tps / 2) >= 2L * ncu * NT_WAVES) CT = 2;
    static const int force_ct = diag_env("PFN_NT_CT") ? atoi(diag_env("PFN_NT_CT")) : 0;   // tuning aid: 1 or 2
    if (force_ct > 0 && CT > 0) CT = std::min(force_ct, (fast && tps >= 2) ? 2 : 1);
    int cshift = 0;
    while (CT > 0 && (CT << cshift) < tps) ++cshift;
    const int rw = NT_WAVES >> cshift;
    dim3 grid(std::min((nrt + rw - 1) / rw, std::max(8, ncu / nslices / 8 * 8)), nslices);

    NtArgs k;
    memset(&k, 0, sizeof(k));
    k.M = a.M; k.ncols = a.ncols; k.ldc = a.ldc;
    k.tps = tps; k.cshift = cshift; k.nq = nq; k.remv = remv;
    k.nrem = nrem;
    k.bias_group = a.bias_group; k.ldr = a.ldr; k.ldg = a.ldg;
    for (int g = 0; g < 8; ++g) k.C[g] = a.C[g];
    k.bias = a.bias; k.rowscale = a.rowscale; k.rowbias = a.rowbias; k.resid = a.resid; k.gate = a.gate;
    k.rng = a.rng; k.rng_stream = a.rng_stream; k.act = a.act; k.p_drop = a.p_drop; k.gate_scale = a.gate_scale;
    k.row0 = a.row0;
    k.reverse = top ? next_sweep_direction() : 0;
    k.c_cm_rows = a.c_cm_rows;
    k.aux_cm_rows = a.aux_cm_rows;

    ProfScope ps(top ? "gemm_nt" : nullptr, bytes, flops, s);
    // ---- large M, the shape of the big-graph launches (every piece K = 129, 129 output columns): full rows per wave, weights
    // streamed through LDS (gemm_nt_ws_kernel) over the largest row range that is WHOLE rounds of the chip (8 row tiles per CU
    // and round; a wave task there is a full 32 x 129 tile, and 6.3 rounds' worth of tiles would cost 7); the remaining rows go
    // to the stationary kernel below, whose tasks are a quarter of that size
    if (top) {   // small M: the K dimension split across the waves of a block (gemm_nt_tiny_kernel)
        static const int tiny_max = diag_env("PFN_NT_TINY_MAX_TILES") ? atoi(diag_env("PFN_NT_TINY_MAX_TILES")) : 256;   // tuning aid; 0 = never
        // (measured, 4-term + 1-term launches of an inference forward, us per launch tiny / stationary: 4 row tiles 5.6 / 15.5,
        //  59 tiles 6.7 / 16.4, 118 tiles 9.1 / 16.8, 236 tiles 15.2 / 17.4, 472 tiles 27.9 / 19.5)
        bool tiny_ok = tiny_max > 0 && nrt <= tiny_max && nq == 4 && remv == 4 && nrem == 1 && a.ngroup == 1 && !a.gate && !a.resid &&
                       (a.act == ACT_NONE || a.act == ACT_RELU) && pieces.size() <= (size_t)TINY_MAX_PIECES && !(a.bias && a.rowscale);
        for (size_t i = 0; i < pieces.size(); ++i) tiny_ok = tiny_ok && pieces[i].klen == KP && last_steps[i] == 1 && pieces[i].group == 0;
        if (tiny_ok) {
            k.npiece = (int)pieces.size();
            for (size_t i = 0; i < pieces.size(); ++i) k.piece[i] = pieces[i];
            gemm_nt_tiny_kernel<<<dim3(nrt, 4), 64 * k.npiece, 0, s>>>(k);
            PFN_CHECK_LAUNCH();
            return PFN_OK;
        }
    }
    if (top) {
        static const int ws_min = diag_env("PFN_NT_WS_MIN_TILES") ? atoi(diag_env("PFN_NT_WS_MIN_TILES")) : 2;   // A/B aid; 0 = never
        const long per_round = (long)ncu * NT_WAVES;
        // Used when the stationary kernel would need MORE THAN TWO LDS slices, i.e. from five 129 x 129 terms (wide.json's K = 6
        // TAGConv: 7 terms -> 32-column slices, every A row read four times): measured at 6470rte x 64 `wide` 36.6 -> 34.6 ms per
        // step.  With <= 4 terms (two slices) the two kernels tie in isolation (4 terms at 414 k rows: 568 vs 582 us) and the
        // streaming one LOSES inside the step (14.2 vs 13.9 ms): its per-piece barrier keeps the 8 waves of a block in lockstep,
        // so all of them run the epilogue (dropout's Philox rounds, the gate loads) at the same time with the matrix pipe idle,
        // where the free-running waves of the stationary kernel overlap one wave's flush with its SIMD partner's MFMAs
        // (profiles/r03_gemm_nt_streaming.txt).
        bool ws_ok = ws_min > 0 && nq == 4 && remv == 4 && nrem == 1 && nslices > 2 && pieces.size() <= (size_t)NT_MAX_PIECES &&
                     (long)nrt >= (long)ws_min * per_round;
        for (size_t i = 0; i < pieces.size(); ++i) ws_ok = ws_ok && pieces[i].klen == KP && last_steps[i] == 1 && pieces[i].kmax >= KP - 12;
        if (ws_ok) {
            const long nround = nrt / per_round;
            const long rows_ws = std::min<long>(a.M, nround * per_round * 32);
            k.npiece = (int)pieces.size();
            for (size_t i = 0; i < pieces.size(); ++i) k.piece[i] = pieces[i];
            k.kuni = KP;
            k.klast = 1;
            k.M = (int)rows_ws;
            const size_t lb = ((size_t)2 * WS_IMG + (size_t)a.ldc) * sizeof(float);
            static std::atomic<uint64_t> lds_raised{0};
            PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_nt_ws_kernel<true, 1>), NT_LDS_BYTES, lds_raised));
            gemm_nt_ws_kernel<true, 1><<<ncu, NT_THREADS, lb, s>>>(k);
            PFN_CHECK_LAUNCH();
            if (rows_ws == a.M) return PFN_OK;
            GemmArgs t = a;                       // the rest of the rows: the same product on offset operands
            t.M = (int)(a.M - rows_ws);
            t.row0 = a.row0 + (int)rows_ws;
            for (int i = 0; i < a.nterm; ++i)   // (chunk-major operand: 4 floats per row inside a plane, the plane stride stays)
                t.term[i].A = a.term[i].A + (size_t)rows_ws * (a.term[i].cm_rows > 0 ? 4 : a.term[i].lda);
            for (int g = 0; g < 8; ++g) t.C[g] = a.C[g] ? a.C[g] + (size_t)rows_ws * (a.c_cm_rows > 0 ? 4 : a.ldc) : nullptr;
            if (a.rowscale) t.rowscale = a.rowscale + rows_ws;
            if (a.resid) t.resid = a.resid + (size_t)rows_ws * (a.aux_cm_rows > 0 ? 4 : a.ldr);
            if (a.gate) t.gate = a.gate + (size_t)rows_ws * (a.aux_cm_rows > 0 ? 4 : a.ldg);
            return launch_gemm_nt_rows(t, s, false);
        }
    }
    if (top) {   // WIDE products (hidden_dim 512: configs/large.json): no trailing column, whole slices of 128 output columns, more
        // weight than the stationary kernel can hold two 32-column quarters of -> the weight-streaming kernel over ALL rows
        // (every piece is a full one: piece-padded images), one launch, every operand row read once per 128 columns
        bool wide_ok = remv == 0 && nq >= 8 && nq % 4 == 0 && pieces.size() <= (size_t)NT_MAX_PIECES && a.ldc == 32 * nq;
        for (size_t i = 0; i < pieces.size(); ++i) wide_ok = wide_ok && pieces[i].klen == KP;
        if (wide_ok) {
            k.npiece = (int)pieces.size();
            for (size_t i = 0; i < pieces.size(); ++i) k.piece[i] = pieces[i];
            k.kuni = KP;
            k.klast = 4;
            const int nsl = nq / 4;
            const int gx = std::max(1, std::min((nrt + NT_WAVES - 1) / NT_WAVES, std::max(1, ncu / nsl)));
            const size_t lb = ((size_t)2 * WS_IMG + (size_t)a.ldc) * sizeof(float);
            static std::atomic<uint64_t> lds_raised_w{0};
            PFN_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_nt_ws_kernel<false, 4>), NT_LDS_BYTES, lds_raised_w));
            gemm_nt_ws_kernel<false, 4><<<dim3(gx, nsl), NT_THREADS, lb, s>>>(k);
            PFN_CHECK_LAUNCH();
            return PFN_OK;
        }
    }
    bool seen[8] = {false, false, false, false, false, false, false, false};
    for (size_t i0 = 0; i0 < pieces.size();) {
        size_t i1 = i0, used = 0;
        while (i1 < pieces.size() && i1 - i0 < (size_t)NT_MAX_PIECES && used + piece_bytes(pieces[i1], tps) <= lds_budget) {
            pieces[i1].lds_off = (int)(used / sizeof(float));
            used += piece_bytes(pieces[i1], tps);
            ++i1;
        }
        if (i1 == i0) {
            set_error("gemm_nt: a %d-row weight piece does not fit in LDS", pieces[i0].klen);
            return PFN_EINVAL;
        }
        k.npiece = (int)(i1 - i0);
        k.bias_lds_off = (int)(used / sizeof(float));
        k.kuni = pieces[i0].klen;
        k.klast = 1;
        for (size_t i = i0; i < i1; ++i) {
            if (pieces[i].klen != k.kuni) k.kuni = 0;
            if (last_steps[i] != 1) k.klast = 4;
        }
        bool here[8] = {false, false, false, false, false, false, false, false};
        for (size_t i = i0; i < i1; ++i) {
            k.piece[i - i0] = pieces[i];
            k.piece[i - i0].gl = pieces[i].group | ((i + 1 == i1 || pieces[i + 1].group != pieces[i].group) ? 256 : 0);
            here[pieces[i].group] = true;
        }
        for (int g = 0; g < 8; ++g) {
            bool later = false;
            for (size_t i = i1; i < pieces.size(); ++i) later |= pieces[i].group == g;
            k.gflags[g] = (seen[g] ? 1 : 0) | (later ? 2 : 0);
            seen[g] |= here[g];
        }
        int rc = PFN_EINVAL;
        const int var = pick_variant(k, CT);
        const size_t lb = used + bias_bytes;
        // pieces summed one by one and added in piece order (gemm_nt_kernel's PAIR): the K = 129 products of three or more terms (a
        // TAGConv's lins, forward and backward) where the kernel has the registers -- one quarter per wave
        static const bool no_pair = diag_env("PFN_NO_NT_PAIR") != nullptr;   // A/B switch: one chain through all terms
        const bool pair = !no_pair && CT == 1 && (var == 0 || var == 1) && k.npiece >= 3;
        if (pair) rc = var == 0 ? launch_variant<1, 0, true>(k, grid, lb, s) : launch_variant<1, 1, true>(k, grid, lb, s);
        // the interleaved flush (gemm_nt_kernel's ILF): two quarters per wave, K = 129, every piece ends a tile, an epilogue of
        // bias / row-scaled bias / ReLU only, whole quarters, every wave owns a tile
        static const bool no_ilf = diag_env("PFN_NO_NT_ILF") != nullptr;   // A/B switch: the flush behind its own multiply
        bool ilf = !no_ilf && CT == 2 && var == 0 && !a.gate && !a.resid && (a.act == ACT_NONE || a.act == ACT_RELU) && tps % 2 == 0 &&
                   nq % tps == 0 && a.ldc == 32 * nq + (remv > 0 ? 4 : 0);
        for (int i = 0; i < k.npiece; ++i) ilf = ilf && (k.piece[i].gl >> 8) != 0;
        for (int g = 0; g < 8; ++g) ilf = ilf && k.gflags[g] == 0;
        if (ilf) rc = launch_variant<2, 0, false, true>(k, grid, lb, s);
#define PFN_NT_CASE(CT_, V_) if (!pair && !ilf && CT == CT_ && var == V_) rc = launch_variant<CT_, V_>(k, grid, lb, s)
        PFN_NT_CASE(0, 0); PFN_NT_CASE(0, 1); PFN_NT_CASE(0, 2); PFN_NT_CASE(0, 3);
        PFN_NT_CASE(1, 0); PFN_NT_CASE(1, 1); PFN_NT_CASE(1, 2); PFN_NT_CASE(1, 3);
        PFN_NT_CASE(2, 0); PFN_NT_CASE(2, 1); PFN_NT_CASE(2, 2);
#undef PFN_NT_CASE
        if (var < 0) set_error("gemm_nt: no CT = %d kernel for this launch (kuni %d, nrem %d)", CT, k.kuni, k.nrem);
        if (rc != PFN_OK) return rc;
        i0 = i1;
    }
    return PFN_OK;
}

}  // namespace pfn

