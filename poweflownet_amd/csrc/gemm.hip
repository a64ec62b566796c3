// fp32 MFMA GEMMs for the per-node dense contractions of the hot path (gfx950).
//
// These carry what the reference runs as torch addmm/mm per EDGE (EdgeAggregation.edge_aggr,
// networks/MPN.py:17-21,:28) and per node (TAGConv.lins, mask_embd :491-495), restructured to per-NODE
// products.  All shapes are "tall-skinny": M = nodes (1e4..1e6), K and N <= a few hundred,
// exact fp32 via v_mfma_f32_16x16x4_f32 (there is no TF32/xf32 on gfx950).
//
//  pack    : once per forward, every weight is copied into zero-padded "LDS images" (tiles of <=132 k-rows x
//            148 floats, one per 144-column block and k chunk, for both orientations W and W^T).  nn.Linear rows
//            of 129 floats are not 16-byte aligned, so this is what makes the weight stream DMA-able.
//  gemm_nt : C = sum_t A_t * B_t (+ epilogue).  One wave owns 16 rows x up to 144 columns (9 accumulator
//            tiles).  A whole k unit of B (<= 132 x 148 floats = 76 KiB) is resident in LDS, filled by
//            global_load_lds (16 B/lane, no VGPR round trip) into the other half of a 2 x 77 KiB ring while
//            the current unit is being multiplied; the wave's A fragment for a unit (9 x float4 per lane,
//            straight from global: rows are private to the wave) is prefetched one unit ahead.  Inside a
//            16-wide k chunk lane group c = lane>>4 supplies k = 16j + 4c + i at MFMA step i, so one 16-byte
//            A load feeds four steps; the 148-float B row stride (= 4 mod 8) makes the matching B reads
//            bank-conflict free.  One barrier per k unit.
//  gemm_tn : weight gradients dW = dY^T X (reduction over the node dimension), 9 waves x (48 x 48) output
//            tiles per block, operands staged through LDS as whole rows with register prefetch of the next
//            stage, split over M and reduced in a second, ordered pass (deterministic; no atomics).  Bias
//            gradients ride along as a virtual ones-column of X.
#include <stdlib.h>

#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = GEMM_CB;           // 144
constexpr int LDB = GEMM_LDB;         // 148: (4 * LDB) % 32 == 16 -> the 4 lane groups hit disjoint banks
constexpr int KC = GEMM_KC;           // 132 k rows per LDS-resident unit
constexpr int NCHUNK = (KC + 15) / 16;            // 9 sixteen-wide k chunks per unit
constexpr int BUF_FLOATS = 77 * 256;              // 77 KiB ring slot (>= KC * LDB floats, whole 1 KiB DMA pieces)
constexpr int ROWS_PER_BLOCK = 64;

// ------------------------------------------------------------------------------------------------ pack
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    // the dropout stream advances once per forward, before any kernel of that forward reads it
    if (a.rng_advance && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.rng_advance[1] += 1;
    const PackJob jb = a.job[blockIdx.y];
    const int ncb = (jb.ld_out + CB - 1) / CB;
    const int K4 = (jb.K + 3) & ~3;
    const long total = (long)ncb * K4 * LDB;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // tiles are laid out [cb][k chunk][row][LDB]; because every chunk but the last has KC rows, the flat
        // index decomposes as cb-major, then absolute k row, then column.
        const int cb = (int)(i / ((long)K4 * LDB));
        const long rem = i - (long)cb * K4 * LDB;
        const int k = (int)(rem / LDB), n = (int)(rem - (long)k * LDB);
        const int gn = cb * CB + n;
        float v = 0.f;
        if (k < jb.K && n < CB && gn < jb.ncols)
            v = jb.trans ? jb.src[(size_t)(jb.wn0 + gn) * jb.ldw + jb.wk0 + k] : jb.src[(size_t)(jb.wk0 + k) * jb.ldw + jb.wn0 + gn];
        jb.dst[i] = v;
    }
}

size_t packed_floats(int K, int ld_out) {
    const int ncb = (ld_out + CB - 1) / CB;
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
// two shapes: CS = 2 -> 8 waves: 4 row groups (16 rows each) x 2 column halves (5 + 4 tiles), 64-row blocks (small M);
//             CS = 1 -> 4 waves x 2 row tiles x all 9 column tiles, 128-row blocks, one wave per SIMD (large M)
constexpr int ZROW_FLOATS = 4 * LDB + 64;       // a zero region every out-of-unit lane reads instead of stale LDS

template <int NWAVES>
__device__ __forceinline__ void dma_unit(const float* __restrict__ src, float* lds_dst, int nbytes, int wave, int lane) {
    // 1 KiB pieces, round-robin over the block's waves; the last piece is clamped to the tile's final 16 bytes for the
    // lanes that would run past it (their LDS bytes land in the slot's unused tail).
    const int npieces = (nbytes + 1023) >> 10;
    const char* base = reinterpret_cast<const char*>(src);
    for (int p = wave; p < npieces; p += NWAVES) {
        int off = (p << 10) + lane * 16;
        off = off < nbytes ? off : nbytes - 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                         (__attribute__((address_space(3))) void*)(lds_dst + (p << 8)), 16, 0, 0);
    }
}

// Multiply one LDS-resident k unit into this wave's RT x NTW accumulator tiles (RT row tiles of 16 rows share every
// B fragment).  B reads of step s+1 are issued before the MFMAs of step s (software pipeline), so together with the
// partner wave on the same SIMD the matrix pipe always has work while an LDS read is in flight.
template <int RT, int NTW, int NTWMAX>
__device__ __forceinline__ void mfma_unit(const float* Bl, const float* zrow_r, const float4 (&a_cur)[RT][NCHUNK], int c,
                                          int rows, int kvalid, f32x4 (&acc)[RT][NTWMAX]) {
#pragma unroll
    for (int j = 0; j < NCHUNK; ++j) {
        const int kleft = kvalid - 16 * j;             // block-uniform: real k's from this chunk on
        if (kleft > 0) {
            // a lane group whose 4 k rows lie beyond the unit reads zeros (rows % 4 == 0: all four or none)
            const float* Bj = (16 * j + 4 * c < rows) ? Bl + 16 * j * LDB : zrow_r;
            float av[RT][4];
#pragma unroll
            for (int q = 0; q < RT; ++q) {
                av[q][0] = a_cur[q][j].x; av[q][1] = a_cur[q][j].y; av[q][2] = a_cur[q][j].z; av[q][3] = a_cur[q][j].w;
            }
            if (kleft >= 4) {
                float b0[NTW], b1[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) b0[t] = Bj[16 * t];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // ping-pong: fetch step i+1 into the other register set, then spend step i
                    if (i < 3) {
#pragma unroll
                        for (int t = 0; t < NTW; ++t) (i & 1 ? b0 : b1)[t] = Bj[(i + 1) * LDB + 16 * t];
                    }
#pragma unroll
                    for (int q = 0; q < RT; ++q)
#pragma unroll
                        for (int t = 0; t < NTW; ++t)
                            acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][i], (i & 1 ? b1 : b0)[t], acc[q][t], 0, 0, 0);
                }
            } else {                                   // ragged tail of the term: 1..3 steps
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (i < kleft) {
                        float b[NTW];
#pragma unroll
                        for (int t = 0; t < NTW; ++t) b[t] = Bj[i * LDB + 16 * t];
#pragma unroll
                        for (int q = 0; q < RT; ++q)
#pragma unroll
                            for (int t = 0; t < NTW; ++t)
                                acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][i], b[t], acc[q][t], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// RT = row tiles per wave: 1 -> 64-row blocks (small problems: more blocks), 2 -> 128-row blocks (large problems: every
// weight byte DMA'd into LDS and every B fragment read from it feeds twice the MFMAs).
template <int RT, int CS>
__global__ __launch_bounds__(CS * 256, 1) void gemm_nt_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x BUF_FLOATS + ZROW_FLOATS
    constexpr int NT_THREADS = CS * 256;
    constexpr int NTWMAX = CS == 2 ? 5 : 9;        // accumulator tiles per wave and row tile
    constexpr int RPB = ROWS_PER_BLOCK * RT;       // rows per block
    constexpr int NIT = (RPB * (CB / 4) + NT_THREADS - 1) / NT_THREADS;   // float4 items per thread in a flush
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, ch = CS == 2 ? wave >> 2 : 0;   // CS = 2: waves w and w+4 share a SIMD (two column halves)
    const int r = lane & 15, c = lane >> 4;
    const int cb = blockIdx.y;
    const int n0 = cb * CB;
    const int tile0 = ch * 5;                      // column tiles [0,5) or [5,9) of the 144-column block
    const int nrb = (a.M + RPB - 1) / RPB;         // row blocks; this block takes bx, bx + gridDim.x, ...
    // tiles this wave really has to produce (narrow outputs leave the second half, or most of the first, idle)
    int ntile_w = (a.ldc - n0 - 16 * tile0 + 15) / 16;
    const int ntile_cap = CS == 2 ? (ch ? 4 : 5) : 9;
    ntile_w = ntile_w < 0 ? 0 : (ntile_w > ntile_cap ? ntile_cap : ntile_w);
    f32x4 acc[RT][NTWMAX];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int t = 0; t < NTWMAX; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* zrow = lds + 2 * BUF_FLOATS;
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
    auto issue = [&](int rb2, int t2, int k2, float4 (&areg)[RT][NCHUNK], float* slot) {
        const GemmTerm& tm = a.term[t2];
        const int K4 = (tm.K + 3) & ~3;
        const int rows = min(KC, K4 - k2 * KC);
        const float* tile = tm.Bp + ((size_t)cb * K4 + (size_t)k2 * KC) * LDB;
        if (!(a.dbg & 1)) dma_unit<NT_THREADS / 64>(tile, slot, rows * LDB * 4, wave, lane);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            const int arow = rb2 * RPB + q * ROWS_PER_BLOCK + rg * 16 + r;
            const bool ok = arow < a.M && ntile_w > 0 && !(a.dbg & 8);
            const float* Arow = tm.A + (size_t)(ok ? arow : 0) * tm.lda + k2 * KC;
#pragma unroll
            for (int j = 0; j < NCHUNK; ++j) {
                const int kk = 16 * j + 4 * c;
                areg[q][j] = (ok && kk < rows) ? *reinterpret_cast<const float4*>(Arow + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    uint64_t seed = 0, offset = 0;
    if (a.act == ACT_DROPOUT_RELU) {
        seed = a.rng[0];
        offset = a.rng[1];
    }
    const float keep_scale = a.act == ACT_DROPOUT_RELU ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int ncol4 = min(CB, a.ldc - n0) >> 2;    // float4 columns of this block's output

    float4 a_cur[RT][NCHUNK], a_nxt[RT][NCHUNK];
    int cur_rb = blockIdx.x, cur_t = -1, cur_k = 0;
    bool have = next_unit(blockIdx.x, -1, 0, cur_rb, cur_t, cur_k);
    if (have) issue(cur_rb, cur_t, cur_k, a_cur, lds);
    __syncthreads();   // waits for the DMA (vmcnt) and publishes the slot
    int slot = 0;
    while (have) {
        int nrb_ = 0, nt_ = 0, nk_ = 0;
        const bool more = next_unit(cur_rb, cur_t, cur_k, nrb_, nt_, nk_);
        if (more) issue(nrb_, nt_, nk_, a_nxt, lds + (slot ^ 1) * BUF_FLOATS);
        // ---- multiply the resident unit
        const int rows = unit_rows(cur_t, cur_k);
        const int kvalid = a.term[cur_t].K - cur_k * KC;   // real (unpadded) k's left in this term
        const float* Bl = lds + slot * BUF_FLOATS + (4 * c) * LDB + r + 16 * tile0;
        if (a.dbg & 2) {
        } else if (ntile_w > 1) {
            if (CS == 1) mfma_unit<RT, 9, NTWMAX>(Bl, zrow + r, a_cur, c, rows, kvalid, acc);
            else if (ch == 0) mfma_unit<RT, 5, NTWMAX>(Bl, zrow + r, a_cur, c, rows, kvalid, acc);
            else mfma_unit<RT, 4, NTWMAX>(Bl, zrow + r, a_cur, c, rows, kvalid, acc);
        } else if (ntile_w == 1) {
            mfma_unit<RT, 1, NTWMAX>(Bl, zrow + r, a_cur, c, rows, kvalid, acc);
        }
        __syncthreads();   // next unit landed (vmcnt(0) precedes the barrier) and this slot is free again
        const int group = a.term[cur_t].group;
        if ((!more || a.term[nt_].group != group || nrb_ != cur_rb) && !(a.dbg & 4)) {
            // ---- flush this group's RPB x 144 tile: accumulators -> LDS (the slot just freed) -> whole-row float4
            // stores with the fused epilogue.  Lane holds D[row = 4c + reg][col = r] of every tile.
            float* stage = lds + slot * BUF_FLOATS;
            const int brow0 = cur_rb * RPB;
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int t = 0; t < NTWMAX; ++t) {
                    if (t < ntile_w) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            stage[(q * ROWS_PER_BLOCK + rg * 16 + 4 * c + reg) * LDB + 16 * (tile0 + t) + r] = acc[q][t][reg];
                    }
                    acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            __syncthreads();
            float* C = a.C[group];
            const bool use_bias = a.bias && (a.bias_group < 0 || a.bias_group == group);
            const float* extra = a.gate ? a.gate : a.resid;
            const int ldx = a.gate ? a.ldg : a.ldr;
            constexpr int FB = 4;                     // items per batch: FB epilogue-operand loads in flight, then FB stores
#pragma unroll 1
            for (int it0 = 0; it0 < NIT; it0 += FB) {
                float4 ex[FB], v4[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int idx = tid + (it0 + u) * NT_THREADS;
                    const int lr = idx / ncol4, q4 = idx - lr * ncol4;
                    const int row = brow0 + lr;
                    const bool ok = lr < RPB && row < a.M;
                    ex[u] = (ok && extra) ? *reinterpret_cast<const float4*>(extra + (size_t)row * ldx + n0 + 4 * q4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                    v4[u] = ok ? *reinterpret_cast<const float4*>(stage + lr * LDB + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int idx = tid + (it0 + u) * NT_THREADS;
                    const int lr = idx / ncol4, q4 = idx - lr * ncol4;
                    const int row = brow0 + lr, col = n0 + 4 * q4;
                    if (lr >= RPB || row >= a.M) continue;
                    float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
                    const float e4[4] = {ex[u].x, ex[u].y, ex[u].z, ex[u].w};
                    const float rs = a.rowscale ? a.rowscale[row] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int cc = col + e;
                        float x = 0.f;
                        if (cc < a.ncols) {
                            x = v[e] + (use_bias ? a.bias[cc] : 0.f);
                            if (a.rowscale) x = fmaf(rs, a.rowbias[cc], x);
                            if (a.resid) x += e4[e];
                            if (a.act == ACT_RELU) {
                                x = fmaxf(x, 0.f);
                            } else if (a.act == ACT_DROPOUT_RELU) {
                                const float uu = uniform_hash(seed, offset, a.rng_stream, (uint64_t)row * a.ncols + cc);
                                x = (uu >= a.p_drop && x > 0.f) ? x * keep_scale : 0.f;
                            }
                            if (a.gate) x = e4[e] > 0.f ? x * a.gate_scale : 0.f;
                        }
                        v[e] = x;
                    }
                    *reinterpret_cast<float4*>(C + (size_t)row * a.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            if (more) {
                // the staged tile must be fully READ before the unit after next is DMA'd over it; the stores
                // themselves may stay in flight (no vmcnt wait: that would stall the next MFMA phase on HBM acks)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int j = 0; j < NCHUNK; ++j) a_cur[q][j] = a_nxt[q][j];
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
    a.ncb = (a.ldc + CB - 1) / CB;
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
    if ((a.gate && a.resid) || (a.gate && a.ldg % 4) || (a.resid && a.ldr % 4)) {
        set_error("gemm_nt: epilogue takes a gate OR a residual, with a row stride that is a multiple of 4");
        return PFN_EINVAL;
    }
    const size_t lds_bytes = (2 * BUF_FLOATS + ZROW_FLOATS) * sizeof(float);
    if (!g_nt_attr_set) {
        PFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<1, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        PFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<2, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        PFN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<2, 2>),
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
    // one 160 KiB-LDS block per CU; persistent blocks stride over the row blocks.  128-row blocks once there are at
    // least two of them per CU (halves the weight DMA and the LDS fragment reads per MFMA), 64-row blocks otherwise.
    const int slots = std::max(1, ncu / a.ncb);
    const int nrb64 = (a.M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    ProfScope ps("gemm_nt", bytes, flops, s);
    static const int wide_rows = getenv("PFN_GEMM_RT2") ? atoi(getenv("PFN_GEMM_RT2")) : 0;   // experiments
    if (wide_rows && nrb64 >= 4 * slots) {
        const int nrb = (a.M + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK);
        if (wide_rows == 1) gemm_nt_kernel<2, 1><<<dim3(std::min(nrb, slots), a.ncb), 256, lds_bytes, s>>>(a);
        else gemm_nt_kernel<2, 2><<<dim3(std::min(nrb, slots), a.ncb), 512, lds_bytes, s>>>(a);
    } else {
        gemm_nt_kernel<1, 2><<<dim3(std::min(nrb64, slots), a.ncb), 512, lds_bytes, s>>>(a);
    }
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ============================================================================================ TN
constexpr int TN_MB = 32;          // node rows per LDS stage
constexpr int TN_LD = CB + 4;      // 148
constexpr int TN_THREADS = 576;    // 9 waves: 3 x 3 macro tiles of 48 x 48
constexpr int TN_MAX_PAIRS = 8;
constexpr int TN_MAX_BLOCKS = 64;  // output macro blocks (144 x 144) per launch
constexpr int TN_Q = CB / 4;       // float4 per staged row (36)

struct TnArgs {
    TnPair pair[TN_MAX_PAIRS];
    int npairs, M, rows_per_split, nsplit, nblocks;
    float* partial;   // [nsplit][nblocks][CB*CB]
    unsigned short blk_pair[TN_MAX_BLOCKS], blk_a0[TN_MAX_BLOCKS], blk_b0[TN_MAX_BLOCKS];
};

// column of the macro block that carries the bias gradient (virtual ones-column), or -1
__device__ __host__ __forceinline__ int tn_bias_col(const TnPair& pr, int b0) {
    if (!pr.bias_out) return -1;
    const int j = pr.nb - b0;              // first column past the real ones
    return (j >= 0 && j < CB) ? j : -1;
}

__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) float ldsA[TN_MB * TN_LD];
    __shared__ __attribute__((aligned(16))) float ldsB[TN_MB * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, c = lane >> 4;
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int wo = wave / 3, wi = wave - 3 * wo;
    const int bcol = tn_bias_col(pr, b0);
    const int na_here = min(CB, pr.na - a0);
    const int nb_here = min(CB, pr.nb - b0) + (bcol >= 0 ? 1 : 0);
    const bool wave_active = (48 * wo < na_here) && (48 * wi < nb_here);
    const int lda4 = (pr.na + 3) & ~3, ldb4 = (pr.nb + 3) & ~3;

    f32x4 acc[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int m_beg = blockIdx.x * a.rows_per_split;
    const int m_end = min(a.M, m_beg + a.rows_per_split);
    // each thread stages two float4 of A and two of B per 32-row stage (32 * 36 = 1152 = 2 * 576)
    float4 pa[2], pb[2];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * TN_THREADS;
            const int m = i / TN_Q, q = i - m * TN_Q;
            const int gm = m0 + m;
            pa[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            pb[h] = pa[h];
            if (gm < m_end) {
                if (a0 + 4 * q < lda4) pa[h] = *reinterpret_cast<const float4*>(pr.A + (size_t)gm * pr.lda + a0 + 4 * q);
                if (b0 + 4 * q < ldb4) pb[h] = *reinterpret_cast<const float4*>(pr.B + (size_t)gm * pr.ldb + b0 + 4 * q);
                if (bcol >= 0 && (bcol >> 2) == q) {
                    const float one = pr.bias_rowscale ? pr.bias_rowscale[gm] : 1.0f;
                    const int bi = bcol & 3;   // selects, not a runtime-indexed store (that would go to scratch)
                    pb[h].x = bi == 0 ? one : pb[h].x;
                    pb[h].y = bi == 1 ? one : pb[h].y;
                    pb[h].z = bi == 2 ? one : pb[h].z;
                    pb[h].w = bi == 3 ? one : pb[h].w;
                }
            }
        }
    };
    if (m_beg < m_end) fetch(m_beg);
    for (int m0 = m_beg; m0 < m_end; m0 += TN_MB) {
        __syncthreads();   // previous stage fully consumed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = tid + h * TN_THREADS;
            const int m = i / TN_Q, q = i - m * TN_Q;
            *reinterpret_cast<float4*>(ldsA + m * TN_LD + 4 * q) = pa[h];
            *reinterpret_cast<float4*>(ldsB + m * TN_LD + 4 * q) = pb[h];
        }
        __syncthreads();
        if (m0 + TN_MB < m_end) fetch(m0 + TN_MB);   // next stage's loads fly under this stage's MFMAs
        if (wave_active) {
#pragma unroll
            for (int g = 0; g < TN_MB / 16; ++g) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int m = 16 * g + 4 * c + s;   // k-permutation over node rows: rows 4 apart per lane group
                    float av[3], bv[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        av[t] = ldsA[m * TN_LD + 48 * wo + 16 * t + r];
                        bv[t] = ldsB[m * TN_LD + 48 * wi + 16 * t + r];
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    float* out = a.partial + ((size_t)blockIdx.x * a.nblocks + by) * (CB * CB);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 48 * wo + 16 * t + 4 * c + reg, j = 48 * wi + 16 * u + r;
                if (i < na_here && j < nb_here) out[i * CB + j] = acc[t][u][reg];
            }
}

// second stage: ordered sum over splits, scatter into the nn.Linear gradient layout (+ bias column)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnArgs a) {
    const int by = blockIdx.y;
    const TnPair pr = a.pair[a.blk_pair[by]];
    const int a0 = a.blk_a0[by], b0 = a.blk_b0[by];
    const int bcol = tn_bias_col(pr, b0);
    const int na_here = min(CB, pr.na - a0);
    const int nb_real = min(CB, pr.nb - b0);
    const int nb_here = nb_real + (bcol >= 0 ? 1 : 0);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= na_here * nb_here) return;
    const int i = idx / nb_here, j = idx - i * nb_here;
    const float* p = a.partial + (size_t)by * (CB * CB) + i * CB + j;
    const size_t stride = (size_t)a.nblocks * (CB * CB);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sp = 0;
    for (; sp + 4 <= a.nsplit; sp += 4) {
        s0 += p[(size_t)sp * stride];
        s1 += p[(size_t)(sp + 1) * stride];
        s2 += p[(size_t)(sp + 2) * stride];
        s3 += p[(size_t)(sp + 3) * stride];
    }
    for (; sp < a.nsplit; ++sp) s0 += p[(size_t)sp * stride];
    const float acc = (s0 + s1) + (s2 + s3);
    if (j < nb_real) pr.G[(size_t)(pr.gn0 + a0 + i) * pr.ldg + pr.gk0 + b0 + j] = acc;
    else pr.bias_out[a0 + i] = acc;
}

// fallback column sums for the (rare) case where the bias column has no room in its macro block
struct ColsumArgs {
    const float* A;
    const float* rowscale;
    float* out;
    int lda, ncols, M;
};
__global__ __launch_bounds__(256) void colsum_fallback_kernel(const ColsumArgs a) {
    __shared__ float red[256];
    const int col = blockIdx.x;
    float acc = 0.f;
    for (int m = threadIdx.x; m < a.M; m += 256) {
        const float v = a.A[(size_t)m * a.lda + col];
        acc += a.rowscale ? a.rowscale[m] * v : v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[col] = red[0];
}

static void tn_split(int64_t M, int nblocks, int& rows_per_split, int& nsplit) {
    // aim for ~one 9-wave block per CU: fewer, longer splits keep the partial-sum traffic (83 KB per block, written
    // then re-read by tn_reduce) below the operand traffic; at least 4 stages (128 rows) per split
    int64_t want = std::max<int64_t>(1, 256 / std::max(1, nblocks));
    int64_t s = std::min<int64_t>(want, (M + 4 * TN_MB - 1) / (4 * TN_MB));
    s = std::max<int64_t>(1, std::min<int64_t>(s, 128));
    rows_per_split = (int)round_up((M + s - 1) / s, TN_MB);
    nsplit = rows_per_split > 0 ? (int)std::max<int64_t>(1, (M + rows_per_split - 1) / rows_per_split) : 1;
}

size_t reduce_ws_floats(int64_t M, int max_na, int max_nb, int max_pairs) {
    (void)M; (void)max_na; (void)max_nb; (void)max_pairs;
    // nsplit * nblocks <= max(512, 64) macro blocks of partials for every split choice of tn_split
    return (size_t)(512 + TN_MAX_BLOCKS) * CB * CB + 1024;
}

int launch_weight_grads(const TnPair* pairs, int npairs, int64_t M, ReduceWs ws, hipStream_t s) {
    int p = 0;
    while (p < npairs) {
        TnArgs ta;
        ta.npairs = 0;
        ta.nblocks = 0;
        ta.M = (int)M;
        ta.partial = ws.partial;
        while (p < npairs && ta.npairs < TN_MAX_PAIRS) {
            const TnPair& pr = pairs[p];
            if (pr.lda % 4 || pr.ldb % 4) {
                set_error("launch_weight_grads: row strides must be multiples of 4");
                return PFN_EINVAL;
            }
            const int nb_blocks = ((pr.na + CB - 1) / CB) * ((pr.nb + CB - 1) / CB);
            if (nb_blocks > TN_MAX_BLOCKS) {
                set_error("launch_weight_grads: a %d x %d weight needs %d macro blocks (> %d)", pr.na, pr.nb, nb_blocks,
                          TN_MAX_BLOCKS);
                return PFN_EINVAL;
            }
            if (ta.nblocks + nb_blocks > TN_MAX_BLOCKS) break;
            for (int a0 = 0; a0 < pr.na; a0 += CB)
                for (int b0 = 0; b0 < pr.nb; b0 += CB) {
                    ta.blk_pair[ta.nblocks] = (unsigned short)ta.npairs;
                    ta.blk_a0[ta.nblocks] = (unsigned short)a0;
                    ta.blk_b0[ta.nblocks] = (unsigned short)b0;
                    ++ta.nblocks;
                }
            ta.pair[ta.npairs++] = pr;
            ++p;
        }
        if (ta.nblocks == 0) continue;
        tn_split(M, ta.nblocks, ta.rows_per_split, ta.nsplit);
        if ((size_t)ta.nsplit * ta.nblocks * CB * CB > ws.floats) {
            set_error("launch_weight_grads: reduction workspace %zu < %zu floats", ws.floats,
                      (size_t)ta.nsplit * ta.nblocks * CB * CB);
            return PFN_ENOSPACE;
        }
        if (M > 0) {
            double flops = 0.0, bytes = 0.0;
            for (int q = 0; q < ta.npairs; ++q) {
                flops += 2.0 * (double)M * ta.pair[q].na * ta.pair[q].nb;
                bytes += 4.0 * (double)M * (ta.pair[q].na + ta.pair[q].nb);
            }
            ProfScope ps("gemm_tn", bytes, flops, s);
            gemm_tn_kernel<<<dim3(ta.nsplit, ta.nblocks), TN_THREADS, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        } else {
            ta.nsplit = 0;
        }
        {
            ProfScope ps("tn_reduce", 0.0, 0.0, s);
            tn_reduce_kernel<<<dim3((CB * (CB + 1) + 255) / 256, ta.nblocks), 256, 0, s>>>(ta);
            PFN_CHECK_LAUNCH();
        }
        for (int q = 0; q < ta.npairs; ++q) {   // bias column without room in its macro block (nb % 144 == 0)
            const TnPair& pr = ta.pair[q];
            if (pr.bias_out && pr.nb % CB == 0) {
                ColsumArgs ca{pr.A, pr.bias_rowscale, pr.bias_out, pr.lda, pr.na, (int)M};
                colsum_fallback_kernel<<<pr.na, 256, 0, s>>>(ca);
                PFN_CHECK_LAUNCH();
            }
        }
    }
    return PFN_OK;
}

}  // namespace pfn
