// Tall-skinny fp32 MFMA GEMM  C = sum_t A_t * B_t (+ fused epilogue)  for the per-node dense contractions of the
// hot path, and the weight re-layout ("pack") that feeds it (gfx950).
//
// Carries what the reference runs as torch addmm/mm per EDGE (EdgeAggregation.edge_aggr, networks/MPN.py:17-21,:28)
// and per node (TAGConv.lins, mask_embd :491-495), restructured to per-NODE products.  M = nodes
// (1e4..1e6), K and N <= a few hundred; exact fp32 on v_mfma_f32_32x32x2_f32 (there is no TF32/xf32 on gfx950).
//
//  pack : once per forward every weight is copied into zero-padded "LDS images": per 128-column block, roundup(K,4)
//         rows of 132 floats (128 main columns + up to 4 "remainder" columns), for both orientations W and W^T.
//         nn.Linear rows of 129 floats are not 16-byte aligned; the images are, which makes the weight stream DMA-able.
//  gemm : block = 64 rows x 128 columns, 8 waves = 2 row groups (32 rows) x 4 column quarters (32 columns): one 32x32
//         accumulator tile (16 VGPRs) per wave, the two waves of a SIMD hide each other's LDS latency.
//         H = 129 = 4*32 + 1: the odd column never gets a tile of its own -- up to 4 trailing output columns are
//         accumulated by VALU dot products from the fragments the wave already holds (B values are LDS broadcasts).
//         A whole k unit of B (<= 132 x 132 floats = 68 KiB) is resident in LDS, filled by global_load_lds (16 B/lane,
//         no VGPR round trip) into the other half of a 2 x 69 KiB ring while the current unit is multiplied; the wave's
//         A fragment for a unit (17 float4 per lane, straight from global: rows are private to a row group) is
//         prefetched one unit ahead.  Inside an 8-wide k chunk lane half kh = lane>>5 supplies k = 8m + 4kh + i at MFMA
//         step i, so one 16-byte A load feeds four steps; B reads are software-pipelined one step ahead.
//         Blocks are persistent (one 160 KiB-LDS block per CU striding over row blocks) and walk ALL output groups of a
//         launch, so DMA and A prefetch stay pipelined across groups and row blocks.  The 32x32 accumulator layout puts
//         32 consecutive columns of a row in one register across lanes -> the epilogue (bias, deg*b2, residual,
//         ReLU / dropout+ReLU, gradient gate) stores whole 128-byte lines straight from registers: no LDS staging, no
//         flush barrier, stores of one wave overlap the MFMAs of the others.
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDB = GEMM_LDB;                   // 132 floats per packed row
constexpr int KC = GEMM_KC;                     // 132 k rows per LDS-resident unit
constexpr int NCH = (KC + 7) / 8;               // 17 eight-wide k chunks per unit
constexpr int SLOT_FLOATS = 69 * 256;           // 69 KiB ring slot (>= KC * LDB floats, whole 1 KiB DMA pieces)
constexpr int ZROW_FLOATS = 4 * LDB;            // zero rows every out-of-unit lane reads instead of stale LDS
constexpr int RPB = 64;                         // rows per block
constexpr int NT_THREADS = 512;

// column plan of an output of `ld` (padded) columns: `remv` trailing columns (0 or 4) go to the VALU path when that
// saves a whole MFMA quarter; `nq` 32-column MFMA quarters; `ncb` 128-column blocks.
__host__ __device__ inline void col_plan(int ld, int& remv, int& nq, int& ncb) {
    const int m = ld & 31;
    remv = (m != 0 && m <= 4) ? m : 0;
    nq = (ld - remv + 31) / 32;
    ncb = nq > 0 ? (nq + 3) / 4 : 1;
}

// ------------------------------------------------------------------------------------------------ pack
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    // the dropout stream advances once per forward, before any kernel of that forward reads it
    if (a.rng_advance && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.rng_advance[1] += 1;
    const PackJob jb = a.job[blockIdx.y];
    int remv, nq, ncb;
    col_plan(jb.ld_out, remv, nq, ncb);
    const int K4 = (jb.K + 3) & ~3;
    const long total = (long)ncb * K4 * LDB;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // [cb][k][LDB]: every k chunk but the last has KC rows, so the flat index is cb-major, then absolute k row
        const int cb = (int)(i / ((long)K4 * LDB));
        const long rem = i - (long)cb * K4 * LDB;
        const int k = (int)(rem / LDB), n = (int)(rem - (long)k * LDB);
        const int gn = cb * GEMM_CB + n;
        float v = 0.f;
        if (k < jb.K && gn < jb.ncols)
            v = jb.trans ? jb.src[(size_t)(jb.wn0 + gn) * jb.ldw + jb.wk0 + k] : jb.src[(size_t)(jb.wk0 + k) * jb.ldw + jb.wn0 + gn];
        jb.dst[i] = v;
    }
}

size_t packed_floats(int K, int ld_out) {
    int remv, nq, ncb;
    col_plan(ld_out, remv, nq, ncb);
    const int K4 = (K + 3) & ~3;
    return (size_t)round_up((int64_t)ncb * K4 * LDB, 256);   // whole KiB: DMA pieces never run off the allocation
}

int launch_pack(const PackJob* jobs, int njobs, uint64_t* rng_advance, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
        PackArgs a;
        a.njobs = std::min(PACK_MAX_JOBS, njobs - j0);
        a.rng_advance = j0 == 0 ? rng_advance : nullptr;
        long biggest = 0;
        for (int j = 0; j < a.njobs; ++j) {
            a.job[j] = jobs[j0 + j];
            biggest = std::max<long>(biggest, (long)packed_floats(jobs[j0 + j].K, jobs[j0 + j].ld_out));
        }
        const int bx = (int)std::min<long>(std::max<long>(1, (biggest + 255) / 256), 64);
        ProfScope ps("pack_weights", 0.0, 0.0, s);
        pack_weights_kernel<<<dim3(bx, a.njobs), 256, 0, s>>>(a);
        PFN_CHECK_LAUNCH();
    }
    return PFN_OK;
}

// ------------------------------------------------------------------------------------------------- NT
__device__ __forceinline__ void dma_unit(const float* __restrict__ src, float* lds_dst, int nbytes, int wave, int lane) {
    // 1 KiB pieces, round-robin over the 8 waves; the last piece is clamped to the tile's final 16 bytes for the
    // lanes that would run past it (their LDS bytes land in the slot's unused tail).
    //
    // Issued as inline asm on purpose: hipcc, when it can see an LDS-DMA in flight, drains vmcnt(0) in front of the next
    // ds_read it cannot disambiguate from the DMA target -- which here is the very first B read of the MFMA phase, i.e.
    // it serialises the weight stream and the A prefetch with the multiply.  Hidden from the compiler, the DMA is
    // waited for by hand (dma_wait) right before the barrier that hands the slot to the readers.
    const int npieces = (nbytes + 1023) >> 10;
    const char* base = reinterpret_cast<const char*>(src);
    for (int p = wave; p < npieces; p += NT_THREADS / 64) {
        int off = (p << 10) + lane * 16;
        off = off < nbytes ? off : nbytes - 16;
        const char* g = base + off;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)(lds_dst + (p << 8))));
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
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// cheap counter-based uniform in [0,1) for the dropout mask (32-bit mixing; the 64-bit state is folded once per launch)
__device__ __forceinline__ float uniform32(uint32_t key0, uint32_t key1, uint32_t idx) {
    uint32_t h = idx * 0x9E3779B1u ^ key0;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16; h += key1;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

struct Epi {   // everything the per-element epilogue needs, resolved once per flush
    const GemmArgs* a;
    bool use_bias;
    uint32_t key0, key1;
    float keep_scale;
};
// `aux` = rowscale[row] (when the GEMM has a row-scaled bias) or gate/resid[row][col] (otherwise), `cb`/`crb` = bias[col] /
// rowbias[col]: all loaded by the caller BEFORE its first store, in one batch (loads cannot be hoisted over the stores
// by the compiler, and one dependent load per element costs a memory latency each).
__device__ __forceinline__ float epilogue(const Epi& e, float v, float aux, float cb, float crb, int row, int col) {
    const GemmArgs& a = *e.a;
    if (col >= a.ncols) return 0.f;
    v += cb;
    if (a.rowscale) v = fmaf(aux, crb, v);
    if (a.resid) v += aux;
    if (a.act == ACT_RELU) {
        v = fmaxf(v, 0.f);
    } else if (a.act == ACT_DROPOUT_RELU) {
        const float u = uniform32(e.key0, e.key1, (uint32_t)row * (uint32_t)a.ncols + (uint32_t)col);
        v = (u >= a.p_drop && v > 0.f) ? v * e.keep_scale : 0.f;
    }
    if (a.gate) v = aux > 0.f ? v * a.gate_scale : 0.f;
    return v;
}

__global__ __launch_bounds__(NT_THREADS, 1) void gemm_nt_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x SLOT_FLOATS + ZROW_FLOATS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rgrp = wave & 1, cq = wave >> 1;     // waves w and w+4 share a SIMD: same rows, column quarters cq, cq+2
    const int r32 = lane & 31, kh = lane >> 5;
    const int cb = blockIdx.y;
    const int n0 = cb * GEMM_CB;
    int remv, nq, ncb;
    col_plan(a.ldc, remv, nq, ncb);
    const bool mfma_on = 4 * cb + cq < nq;                       // this wave owns a live 32-column quarter
    const bool rem_on = cq == 0 && cb == ncb - 1 && remv > 0;    // ... and/or the trailing VALU columns
    const int rem_col = 32 * nq;                                 // first trailing column (global); local = rem_col - n0
    const int nrb = (a.M + RPB - 1) / RPB;         // row blocks; this block takes bx, bx + gridDim.x, ...
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float racc[4] = {0.f, 0.f, 0.f, 0.f};
    float* zrow = lds + 2 * SLOT_FLOATS;
    for (int i = tid; i < ZROW_FLOATS; i += NT_THREADS) zrow[i] = 0.f;

    // ---- unit iterator over (row block, term, k chunk).  The block is persistent: it walks its row blocks and,
    // inside each, ALL output groups (terms arrive sorted by group), flushing the accumulators whenever the group
    // or the row block changes -- so the weight DMA and the A prefetch stay pipelined across groups and row blocks.
    auto nkc_of = [&](int t2) { return (((a.term[t2].K + 3) & ~3) + KC - 1) / KC; };
    auto next_unit = [&](int rb1, int t1, int k1, int& o_rb, int& o_ti, int& o_kc) -> bool {
        int rb2 = rb1, t2 = t1, k2 = k1 + 1;
        if (t2 < 0 || k2 >= nkc_of(t2)) {
            k2 = 0;
            ++t2;
            if (t2 >= a.nterm) {
                t2 = 0;
                rb2 += gridDim.x;
            }
        }
        if (rb2 >= nrb) return false;
        o_rb = rb2;
        o_ti = t2;
        o_kc = k2;
        return true;
    };
    auto unit_rows = [&](int t2, int k2) { const int K4 = (a.term[t2].K + 3) & ~3; return min(KC, K4 - k2 * KC); };
    auto issue = [&](int rb2, int t2, int k2, f32x4 (&areg)[NCH], float* slot) {
        const GemmTerm& tm = a.term[t2];
        const int K4 = (tm.K + 3) & ~3;
        const int rows = min(KC, K4 - k2 * KC);
        const float* tile = tm.Bp + ((size_t)cb * K4 + (size_t)k2 * KC) * LDB;
        if (!(a.dbg & 1)) dma_unit(tile, slot, rows * LDB * 4, wave, lane);
        // The A prefetch is hidden from the compiler as well (asm loads; the hand-over after the barrier names every
        // destination register): hipcc otherwise rotates a_cur/a_nxt through a 2x-unrolled loop and, not knowing about
        // dma_wait(), re-waits on the prefetch with counted vmcnt in the middle of the multiply.  Loads are unconditional
        // from clamped (always valid) addresses: rows past M are clamped to the last row (their results are never
        // stored); k past the row is clamped into the row and zeroed at use time.
        int arow = rb2 * RPB + rgrp * 32 + r32;
        arow = arow < a.M ? arow : a.M - 1;
        const float* Arow = tm.A + (size_t)arow * tm.lda;
        const int kmax = tm.lda - 4;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            int kk = k2 * KC + 8 * m + 4 * kh;
            kk = kk < kmax ? kk : kmax;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[m]) : "v"(Arow + kk) : "memory");
        }
    };

    Epi ep;
    ep.a = &a;
    ep.key0 = ep.key1 = 0;
    ep.keep_scale = 1.f;
    if (a.act == ACT_DROPOUT_RELU) {
        const uint64_t seed = a.rng[0], offset = a.rng[1];
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset * 0x100000001B3ull + ((uint64_t)a.rng_stream << 40) + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        ep.key0 = (uint32_t)z;
        ep.key1 = (uint32_t)(z >> 32);
        ep.keep_scale = 1.0f / (1.0f - a.p_drop);
    }

    f32x4 a_cur[NCH], a_nxt[NCH];
    int cur_rb = blockIdx.x, cur_t = -1, cur_k = 0;
    bool have = next_unit(blockIdx.x, -1, 0, cur_rb, cur_t, cur_k);
    if (have) issue(cur_rb, cur_t, cur_k, a_cur, lds);
    dma_wait();
#pragma unroll
    for (int m = 0; m < NCH; ++m) asm volatile("" : "+v"(a_cur[m]));   // loaded values become visible to the compiler here
    __syncthreads();   // publishes the slot (every wave has waited for its own DMA pieces)
    int slot = 0;
    while (have) {
        int nrb_ = 0, nt_ = 0, nk_ = 0;
        const bool more = next_unit(cur_rb, cur_t, cur_k, nrb_, nt_, nk_);
        if (more) issue(nrb_, nt_, nk_, a_nxt, lds + (slot ^ 1) * SLOT_FLOATS);
        // ---- multiply the resident unit
        const int rows = unit_rows(cur_t, cur_k);
        const int kvalid = a.term[cur_t].K - cur_k * KC;   // real (unpadded) k's left in this term
        const float* S = lds + slot * SLOT_FLOATS;
        if (!(a.dbg & 2)) {
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const int kleft = kvalid - 8 * m;              // block-uniform: real k's from this chunk on
                if (kleft > 0) {
                    // a lane half whose 4 k rows lie beyond the unit reads zeros (rows % 4 == 0: all four or none)
                    const bool lane_in = 8 * m + 4 * kh < rows;
                    const float av[4] = {lane_in ? a_cur[m][0] : 0.f, lane_in ? a_cur[m][1] : 0.f, lane_in ? a_cur[m][2] : 0.f,
                                         lane_in ? a_cur[m][3] : 0.f};
                    if (mfma_on) {
                        const float* Bj = lane_in ? S + (8 * m + 4 * kh) * LDB + 32 * cq + r32 : zrow + r32;
                        float b[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) b[i] = Bj[i * LDB];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < kleft) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b[i], acc, 0, 0, 0);
                    }
                    if (rem_on) {   // trailing columns: lane-local partial dot products, B values are LDS broadcasts
                        const float* Rj = lane_in ? S + (8 * m + 4 * kh) * LDB + (rem_col - n0) : zrow;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float4 rb4 = *reinterpret_cast<const float4*>(Rj + i * LDB);
                            racc[0] = fmaf(av[i], rb4.x, racc[0]);
                            racc[1] = fmaf(av[i], rb4.y, racc[1]);
                            racc[2] = fmaf(av[i], rb4.z, racc[2]);
                            racc[3] = fmaf(av[i], rb4.w, racc[3]);
                        }
                    }
                }
            }
        }
        const int group = a.term[cur_t].group;
        if ((!more || a.term[nt_].group != group || nrb_ != cur_rb) && !(a.dbg & 4)) {
            // ---- flush straight from registers: acc[q] of lane (r32, kh) is D[row (q&3) + 8 (q>>2) + 4 kh][col r32]
            float* C = a.C[group];
            ep.use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
            const int rbase = cur_rb * RPB + rgrp * 32;
            const float* extra = a.gate ? a.gate : a.resid;       // at most one of rowscale / gate / resid per GEMM
            const int ldx = a.gate ? a.ldg : a.ldr;
            if (mfma_on) {
                const int col = n0 + 32 * cq + r32;
                if (col < a.ldc) {
                    const bool real = col < a.ncols;
                    const float cbias = (real && ep.use_bias) ? a.bias[col] : 0.f;
                    const float crb = (real && a.rowscale) ? a.rowbias[col] : 0.f;
                    float aux[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = rbase + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        const int rowc = row < a.M ? row : a.M - 1;
                        aux[q] = a.rowscale ? a.rowscale[rowc] : ((extra && real) ? extra[(size_t)rowc * ldx + col] : 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = rbase + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        if (row < a.M) C[(size_t)row * a.ldc + col] = epilogue(ep, acc[q], aux[q], cbias, crb, row, col);
                    }
                }
            }
            if (rem_on) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = racc[e] + __shfl_xor(racc[e], 32);   // the two k halves
                const int row = rbase + r32;
                if (kh == 0 && row < a.M) {
                    float aux[4], cbias[4], crb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = rem_col + e;
                        const bool real = col < a.ncols;
                        cbias[e] = (real && ep.use_bias) ? a.bias[col] : 0.f;
                        crb[e] = (real && a.rowscale) ? a.rowbias[col] : 0.f;
                        aux[e] = a.rowscale ? a.rowscale[row] : ((extra && real) ? extra[(size_t)row * ldx + col] : 0.f);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = epilogue(ep, v[e], aux[e], cbias[e], crb[e], row, rem_col + e);
                    *reinterpret_cast<float4*>(C + (size_t)row * a.ldc + rem_col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (!more || a.term[nt_].group != group || nrb_ != cur_rb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
        }
        dma_wait();        // this wave's pieces of the next unit have landed ...
        __syncthreads();   // ... so after the barrier the whole next unit is readable and this slot may be refilled
        if (more) {
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                // hand-over AFTER dma_wait() + barrier: the asm-loaded registers become ordinary values here
                asm volatile("" : "+v"(a_nxt[m]));
                a_cur[m] = a_nxt[m];
            }
            cur_rb = nrb_; cur_t = nt_; cur_k = nk_;
            slot ^= 1;
        }
        have = more;
    }
}

static bool g_nt_attr_set = false;

int launch_gemm_nt(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M == 0) return PFN_OK;
    GemmArgs a = a_in;
    int remv, nq;
    col_plan(a.ldc, remv, nq, a.ncb);
    double flops = 0.0, bytes = (double)a.ngroup * a.M * a.ncols * 4.0;
    for (int t = 0; t < a.nterm; ++t) {
        if (a.term[t].lda % 4 != 0 || a.term[t].lda < ((a.term[t].K + 3) & ~3)) {
            set_error("gemm_nt: operand row stride %d must be a multiple of 4 and >= roundup(K=%d, 4)", a.term[t].lda,
                      a.term[t].K);
            return PFN_EINVAL;
        }
        if (a.term[t].Bp == nullptr) {
            set_error("gemm_nt: term %d has no packed weight", t);
            return PFN_EINVAL;
        }
        if (t > 0 && a.term[t].group < a.term[t - 1].group) {
            set_error("gemm_nt: terms must be sorted by output group");
            return PFN_EINVAL;
        }
        flops += 2.0 * a.M * a.term[t].K * a.ncols;
        bytes += (double)a.M * a.term[t].K * 4.0;
    }
    if (a.ldc % 4) {
        set_error("gemm_nt: output row stride %d must be a multiple of 4", a.ldc);
        return PFN_EINVAL;
    }
    const size_t lds_bytes = (2 * SLOT_FLOATS + ZROW_FLOATS) * sizeof(float);
    if (!g_nt_attr_set) {
        PFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        g_nt_attr_set = true;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    static const int dbg = getenv("PFN_GEMM_DBG") ? atoi(getenv("PFN_GEMM_DBG")) : 0;   // timing dissection only
    a.dbg = dbg;
    // one 160 KiB-LDS block per CU; persistent blocks stride over the 64-row blocks
    const int nrb = (a.M + RPB - 1) / RPB;
    dim3 grid(std::min(nrb, std::max(1, ncu / a.ncb)), a.ncb);
    ProfScope ps("gemm_nt", bytes, flops, s);
    gemm_nt_kernel<<<grid, NT_THREADS, lds_bytes, s>>>(a);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // namespace pfn
