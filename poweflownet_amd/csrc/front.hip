// The 4-wide front of the network in ONE launch per direction (gfx950).
//
// MaskEmbdMultiMPN.forward starts with three per-node products whose short side is the 4 node features
// (networks/MPN.py:537 and the first EdgeAggregation's Linear1, :17-21,:28 restructured per node):
//     me_h = relu(mask Wa^T + ba)             (N x H)      mask_embd[0], [1]
//     x0   = x + me_h Wb^T + bb               (N x 4)      mask_embd[2] + the residual add
//     P0   = x0 W1[:, 0:4]^T + b1             (N x H)      layer 0, target half of Linear1
//     Q0   = x0 W1[:, 4:8]^T                  (N x H)      layer 0, source half
// and the backward pass ends with their mirror image
//     g0   = dP0 W1[:, 0:4] + dQ0 W1[:, 4:8]  (N x 4)      gradient w.r.t. x0 (= w.r.t. data.x)
//     dh   = (g0 Wb) * [me_h > 0]             (N x H)      gradient w.r.t. mask_embd's hidden layer.
// As tall-skinny MFMA GEMMs (gemm_nt.hip) these are five launches of ~8-11 us each at case118v2 x 128 -- K = 4 or N = 4
// leaves the matrix cores idle and every launch pays its fixed cost.  Here a thread owns (node row, 4 hidden units): the
// H-wide outputs are 4 + 8 FMAs per element straight from registers, the 4-wide ones a dot product over H reduced through
// LDS in a fixed order (deterministic).  ~30 MFLOP in total: one launch of a few microseconds per direction.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "pfn_internal.hpp"

namespace pfn {

static int wave_max_rows() { return 32768; }   // the row-per-wave kernels serve batches up to this many rows
// 256-thread blocks per CU of the row-per-wave kernels (which: 0 forward, 1 backward, 2 lin_out4).  Every wave first loads its
// lanes' slices of the weights (72 values per lane in the forward), so FEWER, longer-lived waves win as long as the CU still
// has enough of them to hide a row's load chain: case118 x 128 (15 k rows) with 8 / 4 / 3 / 2 / 1 blocks per CU: forward
// 22.5 / 18.1 / 17.1 / 17.0 / 22.4 us, backward 11.9 / 9.7 / 9.8 / 10.5 / - us, lin_out4 7.3 / 7.6 / 6.5 / 8.1 / - us.
static int wave_blocks_per_cu(int which) {
    (void)which;
    return 3;
}

__device__ __forceinline__ float4 ld4f(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4f(float* p, float4 v) { st4_wt(p, v); }   // (every store of this file is a kernel output)

// block = rows_pb rows x nchunk chunk-lanes (nchunk = ld / 4), PERSISTENT over row groups: a thread keeps ONE chunk of four
// hidden units for every row it visits, so its slices of all five weights live in registers for the whole kernel (as
// per-element global loads they made the kernel 3x slower than its memory traffic).  LDS: part[rows_pb][nchunk] | vec[rows_pb].
__device__ __forceinline__ void front_fwd_body(const FrontFwdArgs& a, int bid, int nblk, int ld, int nchunk, int rows_pb,
                                               float4* fl) {
    float4* part = fl;
    float4* vec = fl + (size_t)rows_pb * nchunk;
    const int n = a.n, h = a.h, ldw1 = a.ldw1;
    const int r = threadIdx.x / nchunk, c = threadIdx.x - r * nchunk;
    const bool lane_on = r < rows_pb;
    // this thread's units 4c .. 4c+3 of every weight (zero past H: the pad columns then come out as exact zeros)
    float rwa[4][4], rba[4], rwb[4][4], rw1[4][8], rb1[4];
    // (unconditional loads from a clamped unit, zeroed by a select afterwards: `ok ? load : 0` compiles to an exec-masked
    //  branch with an immediate wait per load -- 72 serialized round trips to L2, 40 us)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * c + i, uc = min(u, h - 1);
        const bool ok = lane_on && u < h;
        const float vba = a.ba[uc], vb1 = a.b1[uc];
        rba[i] = ok ? vba : 0.f;
        rb1[i] = ok ? vb1 : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float va = a.wa[(size_t)uc * 4 + f], vb = a.wb[(size_t)f * h + uc];
            rwa[i][f] = ok ? va : 0.f;
            rwb[i][f] = ok ? vb : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float v1 = a.w1[(size_t)uc * ldw1 + f];
            rw1[i][f] = ok ? v1 : 0.f;
        }
    }
    const float4 bb4 = make_float4(a.bb[0], a.bb[1], a.bb[2], a.bb[3]);
    for (int row0 = bid * rows_pb; row0 < n; row0 += nblk * rows_pb) {
        const int row = row0 + r;
        const bool on = lane_on && row < n;
        if (on) {
            float4 m;                                   // pred_mask.float() (networks/MPN.py:533)
            if (a.mask_dtype == 0) {
                const int64_t* mp = static_cast<const int64_t*>(a.mask) + (size_t)row * 4;
                m = make_float4((float)mp[0], (float)mp[1], (float)mp[2], (float)mp[3]);
            } else {
                m = ld4f(static_cast<const float*>(a.mask) + (size_t)row * 4);
            }
            if (c == 0) st4f(a.maskf + (size_t)row * 4, m);
            float hv[4];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // this chunk's share of me_h Wb^T
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = rba[i];
                v = fmaf(rwa[i][0], m.x, v); v = fmaf(rwa[i][1], m.y, v); v = fmaf(rwa[i][2], m.z, v); v = fmaf(rwa[i][3], m.w, v);
                v = fmaxf(v, 0.f);
                hv[i] = v;
                acc.x = fmaf(rwb[i][0], v, acc.x); acc.y = fmaf(rwb[i][1], v, acc.y);
                acc.z = fmaf(rwb[i][2], v, acc.z); acc.w = fmaf(rwb[i][3], v, acc.w);
            }
            if (a.me_h) st4f(a.me_h + (size_t)row * ld + 4 * c, make_float4(hv[0], hv[1], hv[2], hv[3]));   // (null: no backward follows)
            part[r * nchunk + c] = acc;
        }
        __syncthreads();
        row_sum(part, r, c, nchunk, on);   // fixed-order sum over the row's chunks, left in part[r * nchunk]
        if (on && c == 0) {
            const float4 s = part[r * nchunk];
            const float4 xi = ld4f(a.x + (size_t)row * 4);   // (requested at the top of the trip instead: no gain at 128 graphs, 4-8 % slower at 2048)
            const float4 o = make_float4(xi.x + (s.x + bb4.x), xi.y + (s.y + bb4.y), xi.z + (s.z + bb4.z), xi.w + (s.w + bb4.w));
            vec[r] = o;
            st4f(a.x0 + (size_t)row * 4, o);
        }
        __syncthreads();
        if (on && a.P) {                            // (P null: the first edge stage forms P | Q rows from x0 itself, edge.hip FLY)
            const float4 v = vec[r];
            float p[4], q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float pa = rb1[i];
                pa = fmaf(rw1[i][0], v.x, pa); pa = fmaf(rw1[i][1], v.y, pa); pa = fmaf(rw1[i][2], v.z, pa); pa = fmaf(rw1[i][3], v.w, pa);
                float qb = 0.f;
                qb = fmaf(rw1[i][4], v.x, qb); qb = fmaf(rw1[i][5], v.y, qb); qb = fmaf(rw1[i][6], v.z, qb); qb = fmaf(rw1[i][7], v.w, qb);
                p[i] = pa;
                q[i] = qb;
            }
            st4f(a.P + (size_t)row * ld + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
            st4f(a.Q + (size_t)row * ld + 4 * c, make_float4(q[0], q[1], q[2], q[3]));
        }
        // (the next trip writes part[] only after every reader of this trip has passed the barriers above)
    }
}

// ---- one ROW PER WAVE (H <= 256: a row's <= 64 four-unit chunks are the lanes of one wave).  The row sums become a fixed xor tree
// of DPP / permute shuffles: no LDS, no barrier, so every wave runs its own chain of loads and 32 of them per CU hide each
// other's latency.  Used for small batches (front_row_per_wave); the block-per-row-group body above remains for wider models
// and large batches.
__device__ __forceinline__ float4 wave_sum4(float4 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off);
        v.y += __shfl_xor(v.y, off);
        v.z += __shfl_xor(v.z, off);
        v.w += __shfl_xor(v.w, off);
    }
    return v;
}
__device__ __forceinline__ void front_fwd_wave_body(const FrontFwdArgs& a, int bid, int nblk, int ld, int nchunk) {
    const int n = a.n, h = a.h, ldw1 = a.ldw1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane;
    const bool lane_on = c < nchunk;
    float rwa[4][4], rba[4], rwb[4][4], rw1[4][8], rb1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // unconditional clamped loads + select (see front_fwd_body)
        const int u = 4 * c + i, uc = min(u, h - 1);
        const bool ok = lane_on && u < h;
        const float vba = a.ba[uc], vb1 = a.b1[uc];
        rba[i] = ok ? vba : 0.f;
        rb1[i] = ok ? vb1 : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float va = a.wa[(size_t)uc * 4 + f], vb = a.wb[(size_t)f * h + uc];
            rwa[i][f] = ok ? va : 0.f;
            rwb[i][f] = ok ? vb : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float v1 = a.w1[(size_t)uc * ldw1 + f];
            rw1[i][f] = ok ? v1 : 0.f;
        }
    }
    const float4 bb4 = make_float4(a.bb[0], a.bb[1], a.bb[2], a.bb[3]);
    const int wpb = blockDim.x >> 6;
    auto load_row = [&](int row, float4& m, float4& xi) {   // pred_mask.float() (networks/MPN.py:533) and x: every lane reads the same bytes
        if (a.mask_dtype == 0) {
            const int64_t* mp = static_cast<const int64_t*>(a.mask) + (size_t)row * 4;
            m = make_float4((float)mp[0], (float)mp[1], (float)mp[2], (float)mp[3]);
        } else {
            m = ld4f(static_cast<const float*>(a.mask) + (size_t)row * 4);
        }
        xi = ld4f(a.x + (size_t)row * 4);
    };
    float4 m_n = make_float4(0.f, 0.f, 0.f, 0.f), xi_n = m_n;
    const int row_first = bid * wpb + wave;
    if (row_first < n) load_row(row_first, m_n, xi_n);
    for (int row = row_first; row < n; row += nblk * wpb) {
        const float4 m = m_n, xi = xi_n;
        if (row + nblk * wpb < n) load_row(row + nblk * wpb, m_n, xi_n);   // the next row's inputs, in flight during this row
        float hv[4];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // this chunk's share of me_h Wb^T (zero in lanes past the row)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rba[i];
            v = fmaf(rwa[i][0], m.x, v); v = fmaf(rwa[i][1], m.y, v); v = fmaf(rwa[i][2], m.z, v); v = fmaf(rwa[i][3], m.w, v);
            v = fmaxf(v, 0.f);
            hv[i] = v;
            acc.x = fmaf(rwb[i][0], v, acc.x); acc.y = fmaf(rwb[i][1], v, acc.y);
            acc.z = fmaf(rwb[i][2], v, acc.z); acc.w = fmaf(rwb[i][3], v, acc.w);
        }
        if (lane_on && a.me_h) st4f(a.me_h + (size_t)row * ld + 4 * c, make_float4(hv[0], hv[1], hv[2], hv[3]));
        const float4 s4 = wave_sum4(acc);            // fixed butterfly: every lane ends with the same, deterministic sum
        const float4 o = make_float4(xi.x + (s4.x + bb4.x), xi.y + (s4.y + bb4.y), xi.z + (s4.z + bb4.z), xi.w + (s4.w + bb4.w));
        if (lane == 0) {
            st4f(a.maskf + (size_t)row * 4, m);
            st4f(a.x0 + (size_t)row * 4, o);
        }
        if (lane_on && a.P) {
            float pv[4], qv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float pa = rb1[i];
                pa = fmaf(rw1[i][0], o.x, pa); pa = fmaf(rw1[i][1], o.y, pa); pa = fmaf(rw1[i][2], o.z, pa); pa = fmaf(rw1[i][3], o.w, pa);
                float qb = 0.f;
                qb = fmaf(rw1[i][4], o.x, qb); qb = fmaf(rw1[i][5], o.y, qb); qb = fmaf(rw1[i][6], o.z, qb); qb = fmaf(rw1[i][7], o.w, qb);
                pv[i] = pa;
                qv[i] = qb;
            }
            st4f(a.P + (size_t)row * ld + 4 * c, make_float4(pv[0], pv[1], pv[2], pv[3]));
            st4f(a.Q + (size_t)row * ld + 4 * c, make_float4(qv[0], qv[1], qv[2], qv[3]));
        }
    }
}

// Inference behind an edge stage that forms P | Q itself (EdgeFwdArgs::x0): nothing H-wide is written (no me_h, no P | Q), the
// front's whole output is 32 bytes per row.  One THREAD per row then: every weight index is uniform across the wave (scalar
// loads, the values ride in SGPRs), no LDS, no barriers, coalesced 16-byte loads and stores.  (The block kernel, whose 33
// chunk-lanes per row exist to make H-wide stores coalesced, took 92 us for 241,664 rows with nothing left to store.)
// x0's four dot products are summed in the BLOCK kernel's order -- per chunk of four units an fma chain from zero, chunk k into
// sub-sum k % 8 in chunk order, the eight sub-sums in order (row_sum) -- so a forward pass under no_grad returns the bits of a
// training forward (which stores me_h and takes the block kernel); eight independent chains also hide the fma latency.
__device__ __forceinline__ void front_fwd_thread_body(const FrontFwdArgs& a, int bid, int nblk, int nchunk) {
    const int n = a.n, h = a.h;
    const float4 bb4 = make_float4(a.bb[0], a.bb[1], a.bb[2], a.bb[3]);
    // (read through the constant address space: the compiler then issues SCALAR loads for these wave-uniform addresses; as plain
    //  global pointers inside a struct it could not prove the weights read-only and emitted ~1,160 vector loads per row)
    typedef const float __attribute__((address_space(4))) * cptr;
    const cptr wa = (cptr)a.wa, ba = (cptr)a.ba, wb = (cptr)a.wb;
    for (int row = bid * 256 + (int)threadIdx.x; row < n; row += nblk * 256) {
        float4 m;                                   // pred_mask.float() (networks/MPN.py:533)
        if (a.mask_dtype == 0) {
            const int64_t* mp = static_cast<const int64_t*>(a.mask) + (size_t)row * 4;
            m = make_float4((float)mp[0], (float)mp[1], (float)mp[2], (float)mp[3]);
        } else {
            m = ld4f(static_cast<const float*>(a.mask) + (size_t)row * 4);
        }
        const float4 xi = ld4f(a.x + (size_t)row * 4);
        float4 sub[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sub[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // (units past H: the block kernel multiplies zeroed weights there, i.e. adds exact zeros -- skipped here)
#define PFN_FRONT_UNIT(u_)                                                                                                      \
    {                                                                                                                           \
        float v = ba[u_];                                                                                                       \
        v = fmaf(wa[4 * (u_)], m.x, v); v = fmaf(wa[4 * (u_) + 1], m.y, v); v = fmaf(wa[4 * (u_) + 2], m.z, v);                 \
        v = fmaf(wa[4 * (u_) + 3], m.w, v);                                                                                     \
        v = fmaxf(v, 0.f);                                                                                                      \
        acc.x = fmaf(wb[u_], v, acc.x); acc.y = fmaf(wb[h + (u_)], v, acc.y);                                                   \
        acc.z = fmaf(wb[2 * h + (u_)], v, acc.z); acc.w = fmaf(wb[3 * h + (u_)], v, acc.w);                                     \
    }
        const int nfull8 = (h >> 2) & ~7;           // chunks that come in branch-free groups of eight whole chunks (a branch per
        int k0 = 0;                                 // unit made every scalar weight load wait for the one before it: 41 us)
        for (; k0 < nfull8; k0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 4; ++i) PFN_FRONT_UNIT(4 * (k0 + j) + i)
                sub[j].x += acc.x; sub[j].y += acc.y; sub[j].z += acc.z; sub[j].w += acc.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {               // the last group: chunks and units may run out (uniform branches)
            const int k = k0 + j;
            if (k >= nchunk) break;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 4 * k + i;
                if (u >= h) break;
                PFN_FRONT_UNIT(u)
            }
            sub[j].x += acc.x; sub[j].y += acc.y; sub[j].z += acc.z; sub[j].w += acc.w;
        }
#undef PFN_FRONT_UNIT
        float4 t = sub[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            if (j >= nchunk) break;
            t.x += sub[j].x; t.y += sub[j].y; t.z += sub[j].z; t.w += sub[j].w;
        }
        st4f(a.maskf + (size_t)row * 4, m);
        st4f(a.x0 + (size_t)row * 4, make_float4(xi.x + (t.x + bb4.x), xi.y + (t.y + bb4.y), xi.z + (t.z + bb4.z), xi.w + (t.w + bb4.w)));
    }
}

// Blocks [0, nb_front) run the front, the rest the weight re-layout jobs of the same forward pass (block p -> job p / pack_bx,
// share p % pack_bx): two independent pieces of work, one launch floor (~5 us) less per step.
// Training beyond the latency regime with P | Q formed in the edge walk: the one H-wide tensor left to write is mask_embd's hidden
// layer (kept for the backward pass).  Its workgroups need no reduction and no barrier -- thread (row group slot, chunk) keeps its
// chunk's slices of Wa and ba in registers and walks rows, a row group's chunks are one contiguous 528-byte store -- while x0 and
// maskf come from front_fwd_thread_body in OTHER workgroups of the same launch (independent work; same fma chains as the block
// kernel, so the same bits).  6470rte x 64: 82 us for what the block kernel did in 159 us of barrier-separated phases.
__device__ __forceinline__ void front_meh_body(const FrontFwdArgs& a, int bid, int nblk, int ld, int nchunk) {
    const int n = a.n, h = a.h;
    const int rows_pb = 256 / nchunk;
    const int r = threadIdx.x / nchunk, c = threadIdx.x - r * nchunk;
    if (r >= rows_pb) return;
    float rwa[4][4], rba[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // (unconditional clamped loads + select, see front_fwd_body)
        const int u = 4 * c + i, uc = min(u, h - 1);
        const bool ok = u < h;
        const float vba = a.ba[uc];
        rba[i] = ok ? vba : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float va = a.wa[(size_t)uc * 4 + f];
            rwa[i][f] = ok ? va : 0.f;
        }
    }
    for (int row = bid * rows_pb + r; row < n; row += nblk * rows_pb) {
        float4 m;
        if (a.mask_dtype == 0) {
            const int64_t* mp = static_cast<const int64_t*>(a.mask) + (size_t)row * 4;
            m = make_float4((float)mp[0], (float)mp[1], (float)mp[2], (float)mp[3]);
        } else {
            m = ld4f(static_cast<const float*>(a.mask) + (size_t)row * 4);
        }
        float hv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rba[i];
            v = fmaf(rwa[i][0], m.x, v); v = fmaf(rwa[i][1], m.y, v); v = fmaf(rwa[i][2], m.z, v); v = fmaf(rwa[i][3], m.w, v);
            hv[i] = fmaxf(v, 0.f);
        }
        st4f(a.me_h + (size_t)row * ld + 4 * c, make_float4(hv[0], hv[1], hv[2], hv[3]));
    }
}

// (one instantiation per front body -- MODE 0: one row per wave, 1: block per row group, 2: one row per thread, 3: one row per
//  thread for x0 in the first rows_pb workgroups + front_meh_body in the rest -- so that each
//  gets its own register allocation: as one kernel the thread body's unrolled groups cost the block body its occupancy)
template <int MODE>
__global__ __launch_bounds__(256) void front_pack_kernel(const FrontFwdArgs f, const PackArgs pa, int nb_front, int pack_bx,
                                                         int ld, int nchunk, int rows_pb) {
    extern __shared__ __attribute__((aligned(16))) float4 fl[];
    // the dropout stream advances once per forward, before any kernel of that forward reads it
    if (pa.rng_advance && blockIdx.x == 0 && threadIdx.x == 0) pa.rng_advance[1] += 1;
    if (pa.stamp && blockIdx.x == 0 && threadIdx.x == 0) *pa.stamp = pa.stamp_value;
    slot_ea_body(pa.slot_ea, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    if ((int)blockIdx.x < nb_front) {
        if (MODE == 3) {
            if ((int)blockIdx.x < rows_pb) front_fwd_thread_body(f, blockIdx.x, rows_pb, nchunk);
            else front_meh_body(f, blockIdx.x - rows_pb, nb_front - rows_pb, ld, nchunk);
        } else if (MODE == 2) front_fwd_thread_body(f, blockIdx.x, nb_front, nchunk);
        else if (MODE == 0) front_fwd_wave_body(f, blockIdx.x, nb_front, ld, nchunk);
        else front_fwd_body(f, blockIdx.x, nb_front, ld, nchunk, rows_pb, fl);
        return;
    }
    const int p = blockIdx.x - nb_front, job = p / pack_bx;
    if (job < pa.njobs) pack_job_body(pa.job[job], p - job * pack_bx, pack_bx);
}

__global__ __launch_bounds__(256) void front_bwd_kernel(int n, int h, int ld, int nchunk, int rows_pb, int ldw1,
                                                        const float* __restrict__ dP, const float* __restrict__ dQ,
                                                        const float* __restrict__ me_h, const float* __restrict__ w1,
                                                        const float* __restrict__ wb, float* __restrict__ g0,
                                                        float* __restrict__ dh) {
    extern __shared__ __attribute__((aligned(16))) float4 fl[];
    float4* part = fl;
    float4* vec = fl + (size_t)rows_pb * nchunk;
    const int r = threadIdx.x / nchunk, c = threadIdx.x - r * nchunk;
    const bool lane_on = r < rows_pb;
    float rwb[4][4], rw1[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // unconditional clamped loads + select (see front_fwd_kernel)
        const int u = 4 * c + i, uc = min(u, h - 1);
        const bool ok = lane_on && u < h;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float vb = wb[(size_t)f * h + uc];
            rwb[i][f] = ok ? vb : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float v1 = w1[(size_t)uc * ldw1 + f];
            rw1[i][f] = ok ? v1 : 0.f;
        }
    }
    for (int row0 = blockIdx.x * rows_pb; row0 < n; row0 += gridDim.x * rows_pb) {
        const int row = row0 + r;
        const bool on = lane_on && row < n;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);   // (requested with dP / dQ, used after the row sum's barriers)
        if (on) y = ld4f(me_h + (size_t)row * ld + 4 * c);
        if (on) {
            const float4 p4 = ld4f(dP + (size_t)row * ld + 4 * c), q4 = ld4f(dQ + (size_t)row * ld + 4 * c);
            const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, qv[4] = {q4.x, q4.y, q4.z, q4.w};
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc.x = fmaf(pv[i], rw1[i][0], acc.x); acc.y = fmaf(pv[i], rw1[i][1], acc.y);
                acc.z = fmaf(pv[i], rw1[i][2], acc.z); acc.w = fmaf(pv[i], rw1[i][3], acc.w);
                acc.x = fmaf(qv[i], rw1[i][4], acc.x); acc.y = fmaf(qv[i], rw1[i][5], acc.y);
                acc.z = fmaf(qv[i], rw1[i][6], acc.z); acc.w = fmaf(qv[i], rw1[i][7], acc.w);
            }
            part[r * nchunk + c] = acc;
        }
        __syncthreads();
        row_sum(part, r, c, nchunk, on);
        if (on && c == 0) {
            const float4 s = part[r * nchunk];
            vec[r] = s;
            st4f(g0 + (size_t)row * 4, s);
        }
        __syncthreads();
        if (on) {
            const float4 g = vec[r];
            const float yv[4] = {y.x, y.y, y.z, y.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = 0.f;
                v = fmaf(g.x, rwb[i][0], v); v = fmaf(g.y, rwb[i][1], v); v = fmaf(g.z, rwb[i][2], v); v = fmaf(g.w, rwb[i][3], v);
                o[i] = yv[i] > 0.f ? v : 0.f;
            }
            st4f(dh + (size_t)row * ld + 4 * c, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// one row per wave (see front_fwd_wave_body)
__global__ __launch_bounds__(256) void front_bwd_wave_kernel(int n, int h, int ld, int nchunk, int ldw1,
                                                             const float* __restrict__ dP, const float* __restrict__ dQ,
                                                             const float* __restrict__ me_h, const float* __restrict__ w1,
                                                             const float* __restrict__ wb, float* __restrict__ g0,
                                                             float* __restrict__ dh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane;
    const bool lane_on = c < nchunk;
    float rwb[4][4], rw1[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * c + i, uc = min(u, h - 1);
        const bool ok = lane_on && u < h;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float vb = wb[(size_t)f * h + uc];
            rwb[i][f] = ok ? vb : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float v1 = w1[(size_t)uc * ldw1 + f];
            rw1[i][f] = ok ? v1 : 0.f;
        }
    }
    const int wpb = blockDim.x >> 6, cc = lane_on ? c : 0;   // (lanes past the row re-read chunk 0: zero weights, nothing stored)
    const int step = gridDim.x * wpb, row_first = blockIdx.x * wpb + wave;
    float4 p_n = make_float4(0.f, 0.f, 0.f, 0.f), q_n = p_n, y_n = p_n;
    if (row_first < n) {
        p_n = ld4f(dP + (size_t)row_first * ld + 4 * cc);
        q_n = ld4f(dQ + (size_t)row_first * ld + 4 * cc);
        y_n = ld4f(me_h + (size_t)row_first * ld + 4 * cc);
    }
    for (int row = row_first; row < n; row += step) {
        const float4 p4 = p_n, q4 = q_n, y = y_n;
        if (row + step < n) {   // the next row's inputs, in flight during this row
            p_n = ld4f(dP + (size_t)(row + step) * ld + 4 * cc);
            q_n = ld4f(dQ + (size_t)(row + step) * ld + 4 * cc);
            y_n = ld4f(me_h + (size_t)(row + step) * ld + 4 * cc);
        }
        const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, qv[4] = {q4.x, q4.y, q4.z, q4.w};
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc.x = fmaf(pv[i], rw1[i][0], acc.x); acc.y = fmaf(pv[i], rw1[i][1], acc.y);
            acc.z = fmaf(pv[i], rw1[i][2], acc.z); acc.w = fmaf(pv[i], rw1[i][3], acc.w);
            acc.x = fmaf(qv[i], rw1[i][4], acc.x); acc.y = fmaf(qv[i], rw1[i][5], acc.y);
            acc.z = fmaf(qv[i], rw1[i][6], acc.z); acc.w = fmaf(qv[i], rw1[i][7], acc.w);
        }
        const float4 g = wave_sum4(acc);
        if (lane == 0) st4f(g0 + (size_t)row * 4, g);
        if (lane_on) {
            const float yv[4] = {y.x, y.y, y.z, y.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = 0.f;
                v = fmaf(g.x, rwb[i][0], v); v = fmaf(g.y, rwb[i][1], v); v = fmaf(g.z, rwb[i][2], v); v = fmaf(g.w, rwb[i][3], v);
                o[i] = yv[i] > 0.f ? v : 0.f;
            }
            st4f(dh + (size_t)row * ld + 4 * c, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// out[row][o] = sum_u S[row][u] W2[o][u] + deg[row] b2[o]  for the network's LAST EdgeAggregation layer (Fo <= 4), one row per wave:
// what remains of that layer's second Linear when the graph-resident kernel formed S (a K = 129 -> N = 4 gemm_nt launch took
// 9.4 us at case118 x 128: the matrix cores have nothing to do there)
__global__ __launch_bounds__(256) void lin_out4_wave_kernel(int n, int h, int ld, int nchunk, int fo, const float* __restrict__ S,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ deg, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane;
    const bool lane_on = c < nchunk;
    float rw[4][4];   // rw[o][i] = W2[o][4c + i]
#pragma unroll
    for (int o = 0; o < 4; ++o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = 4 * c + i;
            const float v = w2[(size_t)min(o, fo - 1) * h + min(u, h - 1)];
            rw[o][i] = (lane_on && o < fo && u < h) ? v : 0.f;
        }
    }
    const float4 bv = make_float4(b2[0], fo > 1 ? b2[1] : 0.f, fo > 2 ? b2[2] : 0.f, fo > 3 ? b2[3] : 0.f);
    const int wpb = blockDim.x >> 6, cc = lane_on ? c : 0;
    for (int row = blockIdx.x * wpb + wave; row < n; row += gridDim.x * wpb) {
        const float4 s4 = ld4f(S + (size_t)row * ld + 4 * cc);
        const float d = deg[row];
        float4 acc;
        acc.x = fmaf(s4.w, rw[0][3], fmaf(s4.z, rw[0][2], fmaf(s4.y, rw[0][1], s4.x * rw[0][0])));
        acc.y = fmaf(s4.w, rw[1][3], fmaf(s4.z, rw[1][2], fmaf(s4.y, rw[1][1], s4.x * rw[1][0])));
        acc.z = fmaf(s4.w, rw[2][3], fmaf(s4.z, rw[2][2], fmaf(s4.y, rw[2][1], s4.x * rw[2][0])));
        acc.w = fmaf(s4.w, rw[3][3], fmaf(s4.z, rw[3][2], fmaf(s4.y, rw[3][1], s4.x * rw[3][0])));
        const float4 t = wave_sum4(acc);
        if (lane == 0)
            st4f(out + (size_t)row * 4, make_float4(fmaf(d, bv.x, t.x), fo > 1 ? fmaf(d, bv.y, t.y) : 0.f,
                                                    fo > 2 ? fmaf(d, bv.z, t.z) : 0.f, fo > 3 ? fmaf(d, bv.w, t.w) : 0.f));
    }
}
bool lin_out4_ok(int h, int fo, int ldo, int n) {
    static const bool off = diag_env("PFN_NO_FUSED_BACK") != nullptr;   // (the last layer's special kernels share one A/B switch)
    return !off && fo >= 1 && fo <= 4 && ldo == 4 && ld_of(h) / 4 <= 64 && n <= wave_max_rows();
}
int launch_lin_out4(int n, int h, int fo, const float* S, const float* w2, const float* b2, const float* deg, float* out,
                    hipStream_t s) {
    if (n == 0) return PFN_OK;
    const int ld = ld_of(h);
    ProfScope ps("lin_out4", 0.0, 0.0, s);
    lin_out4_wave_kernel<<<std::min((n + 3) / 4, wave_blocks_per_cu(2) * device_cus()), 256, 0, s>>>(n, h, ld, ld / 4, fo, S, w2, b2, deg, out);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

bool front_fused_ok(int f0, int h) {
    static const bool off = diag_env("PFN_NO_FUSED_FRONT") != nullptr;   // experiments / tests of the generic GEMM path
    return !off && f0 == 4 && ld_of(h) / 4 <= 256;
}

// One row per wave pays in the latency regime only (case118 x 128 = 15 k rows: 25.9 -> 21.4 us forward); with hundreds of
// thousands of rows the half-empty waves (33 of 64 lanes at H = 129) cost more than the barriers of the block version
// (6470rte x 64: backward 182 -> 247 us, measured), which stays for those sizes.
static bool front_row_per_wave(int nchunk, int n) {
    static const bool off = diag_env("PFN_FRONT_BLOCK_ROWS") != nullptr;   // A/B switch: the block-per-row-group kernels
    return !off && nchunk <= 64 && n <= wave_max_rows();
}
// Few rows (the latency regime, where one row per wave serves): the front keeps writing the first layer's P | Q and the edge walk
// gathers them -- forming them in the walk cost more than it saved there (118 rows: edge stage 2.0 -> 2.8 us; 15 k rows: 8.0 -> 12.3 us)
bool front_latency_regime(int h, int n) { return front_row_per_wave(ld_of(h) / 4, n); }
static void front_shape(int h, int& ld, int& nchunk, int& rows_pb, size_t& lds) {
    ld = ld_of(h);
    nchunk = ld / 4;
    rows_pb = 256 / nchunk;
    lds = ((size_t)rows_pb * nchunk + rows_pb) * sizeof(float4);
}

int launch_front_fwd_pack(const FrontFwdArgs& f, const PackJob* jobs, int njobs, uint64_t* rng_advance, hipStream_t s,
                          const SlotEa* slot_ea, int* stamp, int stamp_value) {
    if (f.mask_dtype != 0 && f.mask_dtype != 1) {
        set_error("pred_mask dtype code %d unsupported (0: int64, 1: float32)", f.mask_dtype);
        return PFN_EINVAL;
    }
    int ld, nchunk, rows_pb;
    size_t lds;
    front_shape(f.h, ld, nchunk, rows_pb, lds);
    PackArgs pa;
    if (slot_ea) pa.slot_ea = *slot_ea;
    pa.njobs = std::min(njobs, PACK_MAX_JOBS);
    pa.rng_advance = rng_advance;
    pa.stamp = stamp;
    pa.stamp_value = stamp_value;
    pa.mask = nullptr;
    pa.maskf = nullptr;
    pa.mask_count = 0;
    pa.mask_dtype = 0;
    long biggest = 0;
    for (int j = 0; j < pa.njobs; ++j) {
        pa.job[j] = jobs[j];
        biggest = std::max<long>(biggest, (long)packed_floats(jobs[j].K, jobs[j].ld_out));
    }
    const int pack_bx = (int)std::min<long>(std::max<long>(1, (biggest + 255) / 256), 64);
    const bool per_wave = front_row_per_wave(nchunk, f.n);
    static const bool no_thread_rows = diag_env("PFN_FRONT_NO_THREAD_ROWS") != nullptr;   // A/B switch
    const bool per_thread = !per_wave && !f.P && !f.me_h && !no_thread_rows;   // (instead of the block kernel: same bits)
    const bool split_meh = !per_wave && !f.P && f.me_h && !no_thread_rows;     // training: x0 per thread + me_h in other workgroups
    if (per_wave || per_thread) {   // one row per wave: four rows per 256-thread block, no LDS
        rows_pb = per_thread ? -1 : 0;
        lds = 0;
    }
    const int rows_per_block = per_thread ? 256 : per_wave ? 4 : rows_pb;
    int nb_front = f.n > 0 ? std::min((f.n + rows_per_block - 1) / rows_per_block, (per_thread ? 16 : per_wave ? wave_blocks_per_cu(0) : 8) * device_cus()) : 0;
    if (split_meh && f.n > 0) {   // rows_pb carries the number of x0 workgroups (they come first: the longer chains start first)
        const int nb_meh = std::min((f.n + rows_pb - 1) / rows_pb, 8 * device_cus());
        rows_pb = std::min((f.n + 255) / 256, 16 * device_cus());
        nb_front = rows_pb + nb_meh;
        lds = 0;
    }
    const int nblocks = nb_front + pack_bx * pa.njobs;
    if (nblocks > 0 || rng_advance || stamp) {
        ProfScope ps("front_fwd+pack", 0.0, 0.0, s);
        if (split_meh) front_pack_kernel<3><<<std::max(1, nblocks), 256, lds, s>>>(f, pa, nb_front, pack_bx, ld, nchunk, rows_pb);
        else if (per_thread) front_pack_kernel<2><<<std::max(1, nblocks), 256, lds, s>>>(f, pa, nb_front, pack_bx, ld, nchunk, rows_pb);
        else if (per_wave) front_pack_kernel<0><<<std::max(1, nblocks), 256, lds, s>>>(f, pa, nb_front, pack_bx, ld, nchunk, rows_pb);
        else front_pack_kernel<1><<<std::max(1, nblocks), 256, lds, s>>>(f, pa, nb_front, pack_bx, ld, nchunk, rows_pb);
        PFN_CHECK_LAUNCH();
    }
    if (njobs > pa.njobs) return launch_pack(jobs + pa.njobs, njobs - pa.njobs, nullptr, s);
    return PFN_OK;
}

// P | Q of the first EdgeAggregation layer written out after the fact, with the front's own fma chains (bit-identical to what the
// front stores and to what the edge stage forms on the fly): for the callers that need the rows in memory when the forward pass
// skipped them -- a backward pass that was asked for edge-attribute gradients, the gate export.  Thread = (row, chunk).
__global__ __launch_bounds__(256) void front_pq_kernel(int n, int h, int ld, int nchunk, int ldw1, const float* __restrict__ x0,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       float* __restrict__ P, float* __restrict__ Q) {
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(item / nchunk), c = (int)(item - (long)row * nchunk);
    if (row >= n) return;
    const float4 v = ld4f(x0 + (size_t)row * 4);
    float p[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * c + i, uc = min(u, h - 1);
        const float* w = w1 + (size_t)uc * ldw1;
        float pa = b1[uc];
        pa = fmaf(w[0], v.x, pa); pa = fmaf(w[1], v.y, pa); pa = fmaf(w[2], v.z, pa); pa = fmaf(w[3], v.w, pa);
        float qb = 0.f;
        qb = fmaf(w[4], v.x, qb); qb = fmaf(w[5], v.y, qb); qb = fmaf(w[6], v.z, qb); qb = fmaf(w[7], v.w, qb);
        p[i] = u < h ? pa : 0.f;
        q[i] = u < h ? qb : 0.f;
    }
    st4f(P + (size_t)row * ld + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
    st4f(Q + (size_t)row * ld + 4 * c, make_float4(q[0], q[1], q[2], q[3]));
}
int launch_front_pq(int n, int h, int ldw1, const float* x0, const float* w1, const float* b1, float* P, float* Q, hipStream_t s) {
    if (n == 0) return PFN_OK;
    const int ld = ld_of(h), nchunk = ld / 4;
    ProfScope ps("front_pq", 0.0, 0.0, s);
    front_pq_kernel<<<(int)(((long)n * nchunk + 255) / 256), 256, 0, s>>>(n, h, ld, nchunk, ldw1, x0, w1, b1, P, Q);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

// ---- the backward front beyond the latency regime when mask_embd's hidden layer was NOT stored (FrontFwdArgs::me_h null in a
// training forward: model.hip front_recomputes_meh).  me_h is four fmas per element away from the 16-byte mask row, and dh, which
// the block kernel above writes, has ONE reader: the weight-gradient pair (dh, maskf).  So this kernel recomputes me_h (the
// forward's chain: the same gate bits), forms g0 and dh as above, and accumulates mask_embd's four weight gradients itself
//     dWb[f][u] += g0[row][f] me_h[row][u]   dbb[f] += g0[row][f]   dWa[u][f] += dh[row][u] mask[row][f]   dba[u] += dh[row][u]
// per thread over the rows it visits, per workgroup over its row slots in slot order, per launch over the workgroups in a
// second kernel in workgroup order (fixed order throughout: deterministic).  Neither me_h nor dh touches memory: the forward front
// writes 32 bytes per row, this kernel reads dP and dQ and writes 16 bytes per row, and two N x H pairs leave the weight-gradient
// launch.  Partial sums per workgroup: [9][ld] floats (dWb f = 0..3, dWa f = 0..3, dba) + dbb[4].
// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL access (vmcnt(0)),
// which would pull the prefetched next row group in front of each of the four barriers of a trip
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// row_sum (pfn_internal.hpp) on that barrier: same two-level order, same result
__device__ __forceinline__ void row_sum_lb(float4* part, int r, int c, int nchunk, bool on) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on && c < 8) {
        for (int k = c; k < nchunk; k += 8) {
            const float4 p = part[r * nchunk + k];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
    }
    lds_barrier();
    if (on && c < 8) part[r * nchunk + c] = s;
    lds_barrier();
    if (on && c == 0) {
        float4 t = part[r * nchunk];
        const int m = nchunk < 8 ? nchunk : 8;
        for (int k = 1; k < m; ++k) {
            const float4 p = part[r * nchunk + k];
            t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        part[r * nchunk] = t;
    }
}
constexpr int FWG_SLOTS = 9;
__host__ __device__ inline int fwg_stride(int ld) { return FWG_SLOTS * ld + 4; }
__global__ __launch_bounds__(256) void front_bwd_wg_kernel(int n, int h, int ld, int nchunk, int rows_pb, int ldw1,
                                                           const float* __restrict__ dP, const float* __restrict__ dQ,
                                                           const float* __restrict__ maskf, const float* __restrict__ w1,
                                                           const float* __restrict__ wa, const float* __restrict__ ba,
                                                           const float* __restrict__ wb, float* __restrict__ g0,
                                                           float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float4 fl[];   // part [rows_pb][nchunk] | vec [rows_pb] | w1 [4][2][nchunk] | wb, wa [4][nchunk] | ba [nchunk]
    float4* part = fl;
    float4* vec = fl + (size_t)rows_pb * nchunk;
    float4* s_w1 = vec + rows_pb;
    float4* s_wb = s_w1 + (size_t)nchunk * 8;
    float4* s_wa = s_wb + (size_t)nchunk * 4;
    float4* s_ba = s_wa + (size_t)nchunk * 4;                       // [nchunk] float4 = the chunk's four biases
    const int r = threadIdx.x / nchunk, c = threadIdx.x - r * nchunk;
    const bool lane_on = r < rows_pb;
    // W1's and Wb's slices live in LDS (read per row group: with them in registers next to the 40 accumulators the kernel held 144
    // VGPRs -- three workgroups per CU, too few loads in flight: 187 us for half the bytes of the 183-us kernel above), Wa's and ba's
    // too.  Zero past H: the pad units then contribute exact zeros.
    for (int k = threadIdx.x; k < nchunk * 4; k += blockDim.x) {
        const int u = k, uc = min(u, h - 1);
        const bool ok = u < h;
        const float* w = w1 + (size_t)uc * ldw1;
        const int kc = k >> 2, ki = k & 3;           // unit k = 4 kc + ki -> plane-major [ki][kc]: a row's chunk-lanes read consecutive float4s
        s_w1[(2 * ki) * nchunk + kc] = ok ? make_float4(w[0], w[1], w[2], w[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_w1[(2 * ki + 1) * nchunk + kc] = ok ? make_float4(w[4], w[5], w[6], w[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_wb[ki * nchunk + kc] = ok ? make_float4(wb[uc], wb[(size_t)h + uc], wb[(size_t)2 * h + uc], wb[(size_t)3 * h + uc]) : make_float4(0.f, 0.f, 0.f, 0.f);
        s_wa[ki * nchunk + kc] = ok ? make_float4(wa[(size_t)uc * 4], wa[(size_t)uc * 4 + 1], wa[(size_t)uc * 4 + 2], wa[(size_t)uc * 4 + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float*>(s_ba)[k] = ok ? ba[uc] : 0.f;
    }
    __syncthreads();
    float aWb[4][4], aWa[4][4], aba[4];   // [f][i]: unit 4c + i
    float4 abb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        aba[f] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) aWb[f][i] = aWa[f][i] = 0.f;
    }
    // (the row group after this one is requested BEFORE this one's barriers: with four workgroups per CU a trip's load -> barrier ->
    //  row sum -> barrier chain is otherwise all the latency hiding there is)
    float4 m_n = make_float4(0.f, 0.f, 0.f, 0.f), p_n = m_n, q_n = m_n;
    {
        const int row = blockIdx.x * rows_pb + r;
        if (lane_on && row < n) {
            m_n = ld4f(maskf + (size_t)row * 4);
            p_n = ld4f(dP + (size_t)row * ld + 4 * c);
            q_n = ld4f(dQ + (size_t)row * ld + 4 * c);
        }
    }
    for (int row0 = blockIdx.x * rows_pb; row0 < n; row0 += gridDim.x * rows_pb) {
        const int row = row0 + r;
        const bool on = lane_on && row < n;
        const float4 m = m_n, p4 = p_n, q4 = q_n;
        {
            const int rown = row + gridDim.x * rows_pb;
            if (lane_on && rown < n) {
                m_n = ld4f(maskf + (size_t)rown * 4);
                p_n = ld4f(dP + (size_t)rown * ld + 4 * c);
                q_n = ld4f(dQ + (size_t)rown * ld + 4 * c);
            }
        }
        if (on) {
            const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, qv[4] = {q4.x, q4.y, q4.z, q4.w};
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 wi = s_w1[(2 * i) * nchunk + c], wj = s_w1[(2 * i + 1) * nchunk + c];
                acc.x = fmaf(pv[i], wi.x, acc.x); acc.y = fmaf(pv[i], wi.y, acc.y);
                acc.z = fmaf(pv[i], wi.z, acc.z); acc.w = fmaf(pv[i], wi.w, acc.w);
                acc.x = fmaf(qv[i], wj.x, acc.x); acc.y = fmaf(qv[i], wj.y, acc.y);
                acc.z = fmaf(qv[i], wj.z, acc.z); acc.w = fmaf(qv[i], wj.w, acc.w);
            }
            part[r * nchunk + c] = acc;
        }
        lds_barrier();
        row_sum_lb(part, r, c, nchunk, on);
        if (on && c == 0) {
            const float4 s = part[r * nchunk];
            vec[r] = s;
            st4f(g0 + (size_t)row * 4, s);
        }
        lds_barrier();
        if (on) {
            const float4 g = vec[r];
            const float gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m.x, m.y, m.z, m.w};
            const float4 b4 = s_ba[c];
            const float rba[4] = {b4.x, b4.y, b4.z, b4.w};
            float rwb[4][4], rwa[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w = s_wb[i * nchunk + c], a4 = s_wa[i * nchunk + c];
                rwb[i][0] = w.x; rwb[i][1] = w.y; rwb[i][2] = w.z; rwb[i][3] = w.w;
                rwa[i][0] = a4.x; rwa[i][1] = a4.y; rwa[i][2] = a4.z; rwa[i][3] = a4.w;
            }
            if (c == 0) { abb.x += g.x; abb.y += g.y; abb.z += g.z; abb.w += g.w; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = rba[i];                    // mask_embd's hidden unit, the forward's chain (front_fwd_body / front_meh_body)
                y = fmaf(rwa[i][0], m.x, y); y = fmaf(rwa[i][1], m.y, y); y = fmaf(rwa[i][2], m.z, y); y = fmaf(rwa[i][3], m.w, y);
                y = fmaxf(y, 0.f);
                float v = 0.f;
                v = fmaf(g.x, rwb[i][0], v); v = fmaf(g.y, rwb[i][1], v); v = fmaf(g.z, rwb[i][2], v); v = fmaf(g.w, rwb[i][3], v);
                const float d = y > 0.f ? v : 0.f;   // dh[row][4c + i]
                aba[i] += d;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    aWb[f][i] = fmaf(gv[f], y, aWb[f][i]);
                    aWa[f][i] = fmaf(d, mv[f], aWa[f][i]);
                }
            }
        }
    }
    // ---- the workgroup's partial sums: over its row slots in slot order, one quantity (a float4 of four units) at a time
    float* mine = partial + (size_t)blockIdx.x * fwg_stride(ld);
#pragma unroll
    for (int slot = 0; slot < FWG_SLOTS + 1; ++slot) {
        __syncthreads();
        if (lane_on) {
            float4 v;
            if (slot < 4) v = make_float4(aWb[slot][0], aWb[slot][1], aWb[slot][2], aWb[slot][3]);
            else if (slot < 8) v = make_float4(aWa[slot - 4][0], aWa[slot - 4][1], aWa[slot - 4][2], aWa[slot - 4][3]);
            else if (slot == 8) v = make_float4(aba[0], aba[1], aba[2], aba[3]);
            else v = abb;
            part[r * nchunk + c] = v;
        }
        __syncthreads();
        if (lane_on && r == 0 && (slot < FWG_SLOTS || c == 0)) {
            float4 t = part[c];
            for (int k = 1; k < rows_pb; ++k) {
                const float4 p = part[k * nchunk + c];
                t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
            }
            st4f(slot < FWG_SLOTS ? mine + slot * ld + 4 * c : mine + FWG_SLOTS * ld, t);
        }
    }
}
// sums the workgroups' partials into the four gradients in a fixed order: thread (e, j) of a 4 x 64 workgroup adds the partials of
// workgroups j, j + 64, ... for element e (64 independent chains per element instead of one of 2,048 dependent loads: 181 us as a
// single chain, 41 us with sixteen), then the 64 sub-sums are added in j order
__global__ __launch_bounds__(256) void front_wgrad_reduce_kernel(int nblk, int h, int ld, const float* __restrict__ partial,
                                                                 float* __restrict__ gwa, float* __restrict__ gba,
                                                                 float* __restrict__ gwb, float* __restrict__ gbb,
                                                                 const int* __restrict__ stamp, int stamp_want) {
    __shared__ float sub[64][5];
    const int el = threadIdx.x & 3, j = threadIdx.x >> 2;
    const int e = blockIdx.x * 4 + el, stride = fwg_stride(ld);
    float sacc = 0.f;
    if (e < stride)
        for (int b = j; b < nblk; b += 64) sacc += partial[(size_t)b * stride + e];
    sub[j][el] = sacc;
    __syncthreads();
    if (j != 0 || e >= stride) return;
    float t = sub[0][el];
    for (int k = 1; k < 64; ++k) t += sub[k][el];
    if (stamp && *stamp != stamp_want) t = __builtin_nanf("");   // (pfn_mpn_backward's workspace guard, model.hip WS_STAMP_TRAIN)
    const int slot = e / ld, u = e - slot * ld;
    if (slot >= FWG_SLOTS) gbb[u] = t;                        // (u < 4: the four floats behind the slots)
    else if (u >= h) return;
    else if (slot < 4) gwb[(size_t)slot * h + u] = t;           // dWb [4][h]
    else if (slot < 8) gwa[(size_t)u * 4 + (slot - 4)] = t;     // dWa [h][4]
    else gba[u] = t;
}
size_t front_bwd_wg_scratch_floats(int n, int h) {
    int ld, nchunk, rows_pb;
    size_t lds;
    front_shape(h, ld, nchunk, rows_pb, lds);
    return (size_t)std::min((n + rows_pb - 1) / rows_pb, 8 * device_cus()) * fwg_stride(ld);
}
int launch_front_bwd_wg(int n, int h, int ldw1, const float* dP, const float* dQ, const float* maskf, const float* w1, const float* wa,
                        const float* ba, const float* wb, float* g0, float* scratch, float* gwa, float* gba, float* gwb, float* gbb,
                        hipStream_t s, const int* stamp, int stamp_want) {
    if (n == 0) return PFN_OK;
    int ld, nchunk, rows_pb;
    size_t lds;
    front_shape(h, ld, nchunk, rows_pb, lds);
    const int nblk = std::min((n + rows_pb - 1) / rows_pb, 8 * device_cus());
    {
        ProfScope ps("front_bwd", 0.0, 0.0, s);
        front_bwd_wg_kernel<<<nblk, 256, lds + (size_t)nchunk * 17 * sizeof(float4), s>>>(n, h, ld, nchunk, rows_pb, ldw1, dP, dQ, maskf, w1, wa, ba, wb, g0, scratch);
        PFN_CHECK_LAUNCH();
    }
    ProfScope ps("front_wgrad_reduce", 0.0, 0.0, s);
    front_wgrad_reduce_kernel<<<(fwg_stride(ld) + 3) / 4, 256, 0, s>>>(nblk, h, ld, scratch, gwa, gba, gwb, gbb, stamp, stamp_want);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}
// mask_embd's hidden layer written out after the fact (the gate export, when the forward did not store it)
__global__ __launch_bounds__(256) void front_meh_kernel(const FrontFwdArgs f, int ld, int nchunk) {
    front_meh_body(f, blockIdx.x, gridDim.x, ld, nchunk);
}
int launch_front_meh(int n, int h, const void* mask, int mask_dtype, const float* wa, const float* ba, float* me_h, hipStream_t s) {
    if (n == 0) return PFN_OK;
    int ld, nchunk, rows_pb;
    size_t lds;
    front_shape(h, ld, nchunk, rows_pb, lds);
    FrontFwdArgs f;
    memset(&f, 0, sizeof(f));
    f.n = n; f.h = h; f.mask_dtype = mask_dtype; f.mask = mask; f.wa = wa; f.ba = ba; f.me_h = me_h;
    ProfScope ps("front_meh", 0.0, 0.0, s);
    front_meh_kernel<<<std::min((n + rows_pb - 1) / rows_pb, 8 * device_cus()), 256, 0, s>>>(f, ld, nchunk);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

int launch_front_bwd(int n, int h, int ldw1, const float* dP, const float* dQ, const float* me_h, const float* w1,
                     const float* wb, float* g0, float* dh, hipStream_t s) {
    if (n == 0) return PFN_OK;
    int ld, nchunk, rows_pb;
    size_t lds;
    front_shape(h, ld, nchunk, rows_pb, lds);
    ProfScope ps("front_bwd", 0.0, 0.0, s);
    if (front_row_per_wave(nchunk, n)) {
        front_bwd_wave_kernel<<<std::min((n + 3) / 4, wave_blocks_per_cu(1) * device_cus()), 256, 0, s>>>(n, h, ld, nchunk, ldw1, dP, dQ, me_h, w1, wb, g0, dh);
        PFN_CHECK_LAUNCH();
        return PFN_OK;
    }
    front_bwd_kernel<<<std::min((n + rows_pb - 1) / rows_pb, 8 * device_cus()), 256, lds, s>>>(n, h, ld, nchunk, rows_pb, ldw1, dP, dQ, me_h, w1, wb, g0, dh);
    PFN_CHECK_LAUNCH();
    return PFN_OK;
}

}  // namespace pfn
