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
    // Inline asm on purpose: while hipcc can see an LDS-DMA in flight it waits vmcnt(0) -- not a counted vmcnt -- for
    // every ordinary load it later needs (here: the epilogue operands), which drains the whole prefetch of the next unit in
    // the middle of a flush (measured: 3 us per flush).  Hidden from the compiler, the DMA is waited for by hand
    // (dma_wait) right before the barrier that hands the slot to its readers; the compiler's own counted waits stay
    // correct because VMEM returns in issue order.
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
    const bool o1 = lane & 1, o2 = lane & 2;
    {   // exchange across lane ^ 1: (v0,v1) and (v2,v3)
        const float s01 = dpp_xor1(o1 ? v[0] : v[1]), s23 = dpp_xor1(o1 ? v[2] : v[3]);
        if (o1) { v[0] = s01; v[2] = s23; } else { v[1] = s01; v[3] = s23; }
    }
    {   // exchange across lane ^ 2: (v0,v2) and (v1,v3)
        const float s02 = dpp_xor2(o2 ? v[0] : v[2]), s13 = dpp_xor2(o2 ? v[1] : v[3]);
        if (o2) { v[0] = s02; v[1] = s13; } else { v[2] = s02; v[3] = s13; }
    }
}

// Per-element epilogue.  Every operand arrives BY VALUE (kernel-uniform scalars live in SGPRs; `aux` = rowscale[row] when
// the GEMM has a row-scaled bias, else gate/resid[row][col]; `cb`/`crb` = bias[col] / rowbias[col], zero when absent):
// reaching them through a pointer to the kernel-argument struct made hipcc emit per-lane waterfall loops -- ~2000
// instructions per flush.  All loads are done by the caller in one batch BEFORE its first store.
struct EpiCfg {
    int ncols, act;
    bool has_rowscale, has_resid, has_gate;
    float p_drop, keep_scale, gate_scale;
    uint32_t key0, key1;
};
__device__ __forceinline__ float epilogue(const EpiCfg c, float v, float aux, float cb, float crb, int row, int col) {
    v += cb;
    if (c.has_rowscale) v = fmaf(aux, crb, v);
    if (c.has_resid) v += aux;
    if (c.act == ACT_RELU) {
        v = fmaxf(v, 0.f);
    } else if (c.act == ACT_DROPOUT_RELU) {
        const float u = uniform32(c.key0, c.key1, (uint32_t)row * (uint32_t)c.ncols + (uint32_t)col);
        v = (u >= c.p_drop && v > 0.f) ? v * c.keep_scale : 0.f;
    }
    if (c.has_gate) v = aux > 0.f ? v * c.gate_scale : 0.f;
    return col < c.ncols ? v : 0.f;
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
    // weight DMA of a unit, and the A fragment loads of a unit (one 16-byte load per 8-wide k chunk).  A loads are
    // unconditional from clamped, always valid addresses (a `cond ? load : 0` select would make the compiler wait for the
    // load on the spot): rows past M are clamped to the last row (their results are never stored), k past the row is
    // clamped into the row (those lanes multiply zero B rows).
    auto issue_dma = [&](int t2, int k2, float* slot) {
        const GemmTerm& tm = a.term[t2];
        const int K4 = (tm.K + 3) & ~3;
        const int rows = min(KC, K4 - k2 * KC);
        const float* tile = tm.Bp + ((size_t)cb * K4 + (size_t)k2 * KC) * LDB;
        if (!(a.dbg & 1)) dma_unit(tile, slot, rows * LDB * 4, wave, lane);
    };
    auto a_row_ptr = [&](int rb2, int t2) -> const float* {
        int arow = rb2 * RPB + rgrp * 32 + r32;
        arow = arow < a.M ? arow : a.M - 1;
        return a.term[t2].A + (size_t)arow * a.term[t2].lda;
    };
    auto load_a = [&](const float* Arow, int kmax, int k0, int m) -> f32x4 {
        int kk = k0 + 8 * m + 4 * kh;
        kk = kk < kmax ? kk : kmax;
        return *reinterpret_cast<const f32x4*>(Arow + kk);
    };

    EpiCfg ep;
    ep.ncols = a.ncols;
    ep.act = a.act;
    ep.has_rowscale = a.rowscale != nullptr;
    ep.has_resid = a.resid != nullptr;
    ep.has_gate = a.gate != nullptr;
    ep.p_drop = a.p_drop;
    ep.gate_scale = a.gate_scale;
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

    f32x4 a_cur[NCH];   // ONE register set: chunk m of the NEXT unit is loaded into a_cur[m] right after chunk m is consumed
    int cur_rb = blockIdx.x, cur_t = -1, cur_k = 0;
    bool have = next_unit(blockIdx.x, -1, 0, cur_rb, cur_t, cur_k);
    if (have) {
        issue_dma(cur_t, cur_k, lds);
        const float* Ar = a_row_ptr(cur_rb, cur_t);
#pragma unroll
        for (int m = 0; m < NCH; ++m) a_cur[m] = load_a(Ar, a.term[cur_t].lda - 4, cur_k * KC, m);
    }
    dma_wait();
    __syncthreads();   // publishes the slot (every wave has waited for its own DMA pieces)
    int slot = 0;
    int unit_no = 0;
#define PFN_STAMP(i) do { if (a.timing && blockIdx.x == 0 && tid == 0 && unit_no < 64) a.timing[unit_no * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    while (have) {
        PFN_STAMP(0);
        int nrb_ = 0, nt_ = 0, nk_ = 0;
        const bool more = next_unit(cur_rb, cur_t, cur_k, nrb_, nt_, nk_);
        const float* nxA = nullptr;
        int nx_kmax = 0, nx_k0 = 0;
        const float* nx_tile = nullptr;
        float* nx_slot = lds + (slot ^ 1) * SLOT_FLOATS;
        int nx_bytes = 0;
        bool dma_done = false;
        if (more) {
            const GemmTerm& tn = a.term[nt_];
            const int K4n = (tn.K + 3) & ~3;
            nx_bytes = min(KC, K4n - nk_ * KC) * LDB * 4;
            nx_tile = tn.Bp + ((size_t)cb * K4n + (size_t)nk_ * KC) * LDB;
            nxA = a_row_ptr(nrb_, nt_);
            nx_kmax = a.term[nt_].lda - 4;
            nx_k0 = nk_ * KC;
        }
        const bool pf = more && !(a.dbg & 8);
        // ---- epilogue operands of the flush that follows this unit (if it ends a group / row block): loaded NOW, so
        // they are older than the A refills issued inside the multiply loop and the flush never waits on the prefetch
        const int group = a.term[cur_t].group;
        const bool flush_after = !more || a.term[nt_].group != group || nrb_ != cur_rb;
        const int rbase = cur_rb * RPB + rgrp * 32;
        const int col0 = n0 + 32 * cq + (r32 & ~3);                 // after the quad transpose: 4 columns per lane
        const bool use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
        const float* extra = a.gate ? a.gate : a.resid;           // at most one of rowscale / gate / resid per GEMM
        const int ldx = a.gate ? a.ldg : a.ldr;
        float4 aux[4], raux;
        float cbias[4], crb[4], rcb[4], rcrb[4];
        if (flush_after) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool real = col0 + e < a.ncols, rreal = rem_col + e < a.ncols;
                cbias[e] = (mfma_on && real && use_bias) ? a.bias[col0 + e] : 0.f;
                crb[e] = (mfma_on && real && a.rowscale) ? a.rowbias[col0 + e] : 0.f;
                rcb[e] = (rem_on && rreal && use_bias) ? a.bias[rem_col + e] : 0.f;
                rcrb[e] = (rem_on && rreal && a.rowscale) ? a.rowbias[rem_col + e] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = rbase + (r32 & 3) + 8 * g + 4 * kh;
                const int rowc = row < a.M ? row : a.M - 1;
                aux[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mfma_on && col0 < a.ldc) {
                    if (a.rowscale) {
                        const float rs = a.rowscale[rowc];
                        aux[g] = make_float4(rs, rs, rs, rs);
                    } else if (extra) {
                        aux[g] = *reinterpret_cast<const float4*>(extra + (size_t)rowc * ldx + col0);
                    }
                }
            }
            raux = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rem_on) {
                const int row = rbase + r32;
                const int rowc = row < a.M ? row : a.M - 1;
                if (a.rowscale) {
                    const float rs = a.rowscale[rowc];
                    raux = make_float4(rs, rs, rs, rs);
                } else if (extra) {
                    raux = *reinterpret_cast<const float4*>(extra + (size_t)rowc * ldx + rem_col);
                }
            }
        }
        // small-K units (the slow generic path below) issue the whole DMA up front; the fast path trickles it
        if (more && !(a.dbg & 1)) {
            dma_unit(nx_tile, nx_slot, nx_bytes, wave, lane);
            dma_done = true;
        }
        PFN_STAMP(1);
        // ---- multiply the resident unit
        const int rows = unit_rows(cur_t, cur_k);
        const int kvalid = a.term[cur_t].K - cur_k * KC;   // real (unpadded) k's left in this term
        const float* S = lds + slot * SLOT_FLOATS;
        if (!(a.dbg & 2)) {
            const int nreal_rem = a.ncols - rem_col;           // real trailing columns (1 for H = 129); the rest is padding
            // chunks whose 8 k's are all real and inside the unit run as straight-line code (no per-step branches, no
            // zero-row redirect): the compiler can then pipeline the LDS reads of later chunks under earlier MFMAs
            const int nfull = min(rows, kvalid) >> 3;
            constexpr int FAST = 16;                           // H = 129: 16 full chunks + one ragged chunk
            int m_done = 0;
            if (nfull >= FAST) {
                const float* Bj = S + (4 * kh) * LDB + 32 * cq + r32;
                const float* Rj = S + (4 * kh) * LDB + (rem_col - n0);
#pragma unroll
                for (int m = 0; m < FAST; ++m) {
                    if (mfma_on) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float bv = (a.dbg & 16) ? 1.0f : Bj[(8 * m + i) * LDB];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][i], bv, acc, 0, 0, 0);
                        }
                    }
                    if (rem_on && !(a.dbg & 64)) {
                        if (nreal_rem <= 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) racc[0] = fmaf(a_cur[m][i], Rj[(8 * m + i) * LDB], racc[0]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float4 rb4 = *reinterpret_cast<const float4*>(Rj + (8 * m + i) * LDB);
                                racc[0] = fmaf(a_cur[m][i], rb4.x, racc[0]);
                                racc[1] = fmaf(a_cur[m][i], rb4.y, racc[1]);
                                racc[2] = fmaf(a_cur[m][i], rb4.z, racc[2]);
                                racc[3] = fmaf(a_cur[m][i], rb4.w, racc[3]);
                            }
                        }
                    }
                    if (pf && !(a.dbg & 32)) a_cur[m] = load_a(nxA, nx_kmax, nx_k0, m);   // chunk m consumed: refill it for the next unit
                }
                m_done = FAST;
            }
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const int kleft = kvalid - 8 * m;              // block-uniform: real k's from this chunk on
                if (m >= m_done && kleft > 0) {
                    // a lane half whose 4 k rows lie beyond the unit reads zeros (rows % 4 == 0: all four or none)
                    const bool lane_in = 8 * m + 4 * kh < rows;
                    const float av[4] = {a_cur[m][0], a_cur[m][1], a_cur[m][2], a_cur[m][3]};
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
                if (m >= m_done && pf) a_cur[m] = load_a(nxA, nx_kmax, nx_k0, m);
            }
        } else if (pf) {
#pragma unroll
            for (int m = 0; m < NCH; ++m) a_cur[m] = load_a(nxA, nx_kmax, nx_k0, m);
        }
        if (more && !dma_done && !(a.dbg & 1)) dma_unit(nx_tile, nx_slot, nx_bytes, wave, lane);
        PFN_STAMP(2);
        // The next unit's DMA pieces (and A refills) have had the whole multiply to land: wait for them HERE, before the
        // flush issues its stores -- vmcnt counts stores too, so a vmcnt(0) in front of the barrier would also wait for
        // the HBM write acknowledgements of the tile just flushed (measured: ~3 us per flush).
        dma_wait();
        if (flush_after && !(a.dbg & 4)) {
            // ---- flush straight from registers: acc[q] of lane (r32, kh) is D[row (q&3) + 8 (q>>2) + 4 kh][col r32];
            // after the quad transpose lane (u = r32 >> 2, j = r32 & 3) holds, for register group g, row
            // rbase + j + 8 g + 4 kh and the four columns col0 .. col0 + 3
            float* C = a.C[group];
            if (mfma_on && col0 < a.ldc) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    quad_transpose(v, lane);
                    const int row = rbase + (r32 & 3) + 8 * g + 4 * kh;
                    const float ax[4] = {aux[g].x, aux[g].y, aux[g].z, aux[g].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = epilogue(ep, v[e], ax[e], cbias[e], crb[e], row, col0 + e);
                    if (row < a.M) *reinterpret_cast<float4*>(C + (size_t)row * a.ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            if (rem_on) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = racc[e] + __shfl_xor(racc[e], 32);   // the two k halves
                const int row = rbase + r32;
                if (kh == 0 && row < a.M) {
                    const float ax[4] = {raux.x, raux.y, raux.z, raux.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = epilogue(ep, v[e], ax[e], rcb[e], rcrb[e], row, rem_col + e);
                    *reinterpret_cast<float4*>(C + (size_t)row * a.ldc + rem_col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (flush_after) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
        }
        PFN_STAMP(3);
        // raw barrier (LDS reads done, no vmcnt drain: the flush's stores stay in flight across it); every wave waited for
        // its own DMA pieces above, so after the barrier the whole next unit is readable and this slot may be refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PFN_STAMP(4);
        ++unit_no;
        if (more) {
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
