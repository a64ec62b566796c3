// MFMA tile building blocks of the graph-resident EdgeAggregation kernels (ea_seg.hip): A fragments in registers, one weight
// quarter in LDS (LDS-DMA), one 32 x 32 fp32 tile per wave.
#pragma once
#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_WAVES = 8;          // waves that share one seg_copy_b (ea_seg.hip: 512-thread blocks)

// One 32 x 32 MFMA tile of  A[rows r_first..][0..K) * image quarter q  (K <= 136: one piece), written to an LDS tile.
//   A  : row-major, lda floats per row, straight from global memory into registers (17 float4 per lane, all requested at once);
//        the lane's row is clamped to r_last (results of clamped rows land in pad rows of the tile)
//   B  : the quarter of the packed image (pack_job_body: group g = k / 4 -> [(q * G + g) * 128 + col * 4 + (k & 3)]), copied
//        ONCE per block into LDS by LDS-DMA and shared by the row-tile waves (as per-lane register fragments every wave pulled its
//        own 17 KB copy through L1: 11 of the kernel's 33 us)
// The k order (chunk m, step i: lane half kh supplies k = 8m + 4kh + i) is gemm_nt's, so the tile sums are bit-identical to it.
constexpr int SG_NCH = 17;           // eight-wide k chunks: K8 <= 136
struct SegA { f32x4 av[SG_NCH]; };
__device__ __forceinline__ void seg_load_a(SegA& t, const float* __restrict__ A, int lda, int K8, int r_first, int r_last, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* arow = A + (size_t)min(r_first + r32, r_last) * lda;
    const int kmax = lda - 4;
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) {
        const int mc = min(m, (K8 >> 3) - 1);                        // clamped: chunks past K8 are loaded but not multiplied
        t.av[m] = *reinterpret_cast<const f32x4*>(arow + min(8 * mc + 4 * kh, kmax));
    }
}
// The same fragment (K8 == 8 * SG_NCH) requested with loads HIDDEN from hipcc and consumed WHILE IT ARRIVES: chunk m is waited
// for by hand right before its MFMAs (seg_wait_chunk<m>: vector-memory reads return in order, so "at most SG_NCH - 1 - m younger
// operations outstanding" means chunk m has landed; any other younger operation only makes the wait stricter).  Protocol of the
// calling kernel -- each step checked in the ISA of every kernel that uses it:
//   1. every OTHER load of the prologue is consumed and a compiler-VISIBLE `s_waitcnt vmcnt(0)` (seg_drain_visible) has run before
//      these loads go out: hipcc then knows of nothing pending and places no wait of its own behind them (with visible loads it
//      cannot prove finished across the kernels' branches it puts a vmcnt(0) -- a full drain -- in front of the first MFMA);
//   2. the barrier that follows waits for LDS only (seg_lds_barrier);
//   3. nothing reads, copies or spills t.av[] before its chunk's wait (no "+v" ties: those made hipcc copy a fragment register
//      BEFORE the wait); the waits are pinned between sched_barriers so no MFMA moves above its wait.
// Why: the fragment -- 16 bytes per lane out of 32 different rows per instruction -- is the slowest thing a graph-resident
// prologue asks for, and requested up front it held the barrier back until its LAST chunk was in; the multiply began only then.
__device__ __forceinline__ void seg_drain_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt untouched
__device__ __forceinline__ void seg_load_a_async(SegA& t, const float* __restrict__ A, int lda, int r_first, int r_last, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* arow = A + (size_t)min(r_first + r32, r_last) * lda;
    const float* p0 = arow + 4 * kh;
    const float* plast = arow + min(8 * (SG_NCH - 1) + 4 * kh, lda - 4);   // (the last chunk's upper half is clamped into the row)
#define PFN_SEG_LD(m) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(t.av[m]) : "v"(p0), "n"(32 * (m)) : "memory")
    PFN_SEG_LD(0); PFN_SEG_LD(1); PFN_SEG_LD(2); PFN_SEG_LD(3); PFN_SEG_LD(4); PFN_SEG_LD(5); PFN_SEG_LD(6); PFN_SEG_LD(7);
    PFN_SEG_LD(8); PFN_SEG_LD(9); PFN_SEG_LD(10); PFN_SEG_LD(11); PFN_SEG_LD(12); PFN_SEG_LD(13); PFN_SEG_LD(14); PFN_SEG_LD(15);
#undef PFN_SEG_LD
    static_assert(SG_NCH == 17, "seg_load_a_async spells out SG_NCH loads");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t.av[SG_NCH - 1]) : "v"(plast) : "memory");
}
//   4. ALL FOUR registers of a chunk stay reserved until its wait, also where the multiply reads one of them (the last chunk of
//      K = 129): hipcc had handed the three "dead" registers of that tuple to an LDS address and a loop counter while the load
//      that overwrites them was in flight -- the empty asm behind the wait names the whole tuple.
template <int M>
__device__ __forceinline__ void seg_wait_chunk(const SegA& t) {
    static_assert(M >= 0 && M < SG_NCH, "chunk index");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SG_NCH - 1 - M) : "memory");
    asm volatile("" ::"v"(t.av[M]));
    __builtin_amdgcn_sched_barrier(0);
}
// One 1 KiB LDS-DMA (64 lanes x 16 bytes; LDS destination = wave-uniform base + lane * 16); inline asm as in gemm_nt.hip: hidden
// from the compiler, waited for by hand (seg_dma_wait) before the barrier that publishes the copy
__device__ __forceinline__ void seg_dma_1k(const char* g, float* lds_dst) {
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
// Barrier that publishes LDS writes only: __syncthreads() also waits (vmcnt(0)) until every global STORE of the wave is acknowledged --
// a round trip of 1-2 us under load that a kernel pays at every barrier that follows its output stores (phase timestamps: 1.5 us
// per hop of seg_lin_hops_kernel).  Use only where no thread reads another thread's GLOBAL writes after the barrier.
__device__ __forceinline__ void seg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void sg_st4_wt(float* p, float4 v) { st4_wt(p, v); }   // (pfn_internal.hpp: write-through output store)
__device__ __forceinline__ void seg_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// quarter q of a packed image (G * 128 floats, whole KiBs) -> LDS, the 1 KiB pieces dealt round-robin to the block's waves
__device__ __forceinline__ void seg_copy_b(float* dst, const float* __restrict__ Bp, int q, int K8, int wave, int lane,
                                           int nwaves = SG_WAVES) {
    const int nfl = (K8 >> 2) * 128;
    const char* src = reinterpret_cast<const char*>(Bp + (size_t)q * nfl) + lane * 16;
    for (int off = wave * 256; off < nfl; off += nwaves * 256) seg_dma_1k(src + (size_t)off * 4, dst + off);
}
// (one accumulator chain: a second, independent one changed nothing -- the MFMA pipe is shared by 2-4 waves per SIMD here, and
//  tools/ubench/mfma_peak.hip reaches 147 TF with a single dependent chain per wave)
template <bool FULL>   // FULL: K8 == 136, all 17 chunks live -> straight-line code (the per-chunk guards cost ~50 SGPRs of branch state)
__device__ __forceinline__ f32x16 seg_mma_t(const SegA& t, const float* bl, int K8, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* bp = bl + kh * 128 + r32 * 4;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    f32x4 b = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
    for (int m = 0; m < SG_NCH; ++m) {
        if (FULL || 8 * m < K8) {
            f32x4 bn = b;
            if (FULL ? m + 1 < SG_NCH : 8 * (m + 1) < K8) bn = *reinterpret_cast<const f32x4*>(bp + (m + 1) * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.av[m][i], b[i], acc, 0, 0, 0);
            b = bn;
        }
    }
    return acc;
}
__device__ __forceinline__ f32x16 seg_mma(const SegA& t, const float* bl, int K8, int lane) {
    return K8 == 8 * SG_NCH ? seg_mma_t<true>(t, bl, K8, lane) : seg_mma_t<false>(t, bl, K8, lane);
}
// the FULL multiply over a fragment requested with seg_load_a_async: every chunk is waited for right before its MFMAs
template <int M>
__device__ __forceinline__ void seg_mma_async_from(f32x16& acc, f32x4& b, const SegA& t, const float* bp) {
    if constexpr (M < SG_NCH) {
        f32x4 bn = b;
        if (M + 1 < SG_NCH) bn = *reinterpret_cast<const f32x4*>(bp + (M + 1) * 256);
        seg_wait_chunk<M>(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.av[M][i], b[i], acc, 0, 0, 0);
        b = bn;
        seg_mma_async_from<M + 1>(acc, b, t, bp);
    }
}
__device__ __forceinline__ f32x16 seg_mma_async(const SegA& t, const float* bl, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const float* bp = bl + kh * 128 + r32 * 4;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    f32x4 b = *reinterpret_cast<const f32x4*>(bp);
    seg_mma_async_from<0>(acc, b, t, bp);
    return acc;
}


// ---- geometry and small helpers shared by the graph-resident kernels (ea_seg.hip, seg_lin_hops.hip)
constexpr int SG_THREADS = 512;
constexpr int SG_TW = 36;            // LDS tile row stride (floats): 32 quarter columns + up to 4 trailing VALU columns
constexpr int SG_MAX_ROWS = 128;     // rows of whole graphs per workgroup (4 row tiles: one MFMA task per wave at most)
constexpr int SG_LDS_BYTES = 78 * 1024;   // two workgroups per CU

__device__ __forceinline__ float4 sg_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sg_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// accumulator register 4g + e of lane (r32, kh) = row 8g + 4kh + e, column r32
__device__ __forceinline__ void seg_store_tile(const f32x16& acc, int q, const float* __restrict__ bias, int ncols, float* tile,
                                               int trow0, int lane) {
    const int r32 = lane & 31, kh = lane >> 5;
    const int col = 32 * q + r32;
    const float cb = (bias && col < ncols) ? bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[(size_t)(trow0 + 8 * (j >> 2) + 4 * kh + (j & 3)) * SG_TW + r32] = acc[j] + cb;
}

// column-slice geometry shared by the kernels: block y = q covers the 32-column quarter q (up to 8 float4 chunks); the LAST
// quarter's block also owns the `remv` trailing columns (one more chunk, LDS tile columns 32..35)
struct SegCols {
    int q, nq, remv, cw, col0;   // cw = float4 chunks of this block's slice
    bool rem;                    // this block owns the trailing columns
};
__device__ __forceinline__ SegCols seg_cols(int ld, int by = -1) {
    SegCols c;
    col_plan(ld, c.remv, c.nq);
    if (by < 0) by = (int)blockIdx.y;
    // (reversed: the last quarter's block, which also owns the trailing columns and is the launch's critical path, is dispatched
    //  first -- the second half of a launch's workgroups reaches its first barrier ~3.6 us later than the first half, measured)
    c.q = c.nq - 1 - by;
    c.rem = c.remv > 0 && c.q == c.nq - 1;
    c.col0 = 32 * c.q;
    c.cw = min(8, (ld - c.remv - c.col0) >> 2) + (c.rem ? 1 : 0);
    return c;
}
// LDS tile column of chunk lc / its global column
__device__ __forceinline__ int seg_tcol(const SegCols& c, int lc) { return (c.rem && lc == c.cw - 1) ? 32 : 4 * lc; }
__device__ __forceinline__ int seg_gcol(const SegCols& c, int lc) { return (c.rem && lc == c.cw - 1) ? 32 * c.nq : c.col0 + 4 * lc; }
// global column of LDS tile column t (0..31: the quarter; 32..35: the trailing columns), -1: not in this block
__device__ __forceinline__ int seg_col_of_tile(const SegCols& c, int t) { return t < 32 ? c.col0 + t : (c.rem && t - 32 < c.remv ? 32 * c.nq + t - 32 : -1); }

__device__ __forceinline__ float4 sg_add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sg_fma4(float a, float4 x, float4 acc) {
    return make_float4(fmaf(a, x.x, acc.x), fmaf(a, x.y, acc.y), fmaf(a, x.z, acc.z), fmaf(a, x.w, acc.w));
}
__device__ __forceinline__ float4 sg_relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }


}  // namespace pfn
