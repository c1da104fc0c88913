// Inverse index ("CSR transpose") of a neighbour table, the common first stage of the two scatter-add
// backward ops (grouping_operation and three_interpolate).
//
// Given idx (B, L) with values in [0, nbins), produce per cloud
//     start (nbins+1) : start[k]..start[k+1] = the slots l with idx[l] == k
//     order (L)       : those slots, bin by bin, ASCENDING slot id inside a bin
// so that a backward pass becomes one gather-sum per destination element: no float atomics, and the
// summation order is fixed => bitwise reproducible (the oracle accumulates in the same ascending order).
//
// One workgroup per cloud, everything in LDS: histogram with LDS integer atomics, block scan, unordered
// atomic fill, then every bin's (short) segment is insertion-sorted by slot id, which erases the only
// timing-dependent part.  Clouds whose tables do not fit LDS use the same algorithm on caller scratch in HBM.
#include "tp3d_common.h"

namespace tp3d {

constexpr int CSR_BLOCK = 1024;
constexpr int CSR_SMALL_BIN = 24;   // bins up to this many slots are insertion-sorted by one thread
constexpr int CSR_RANK_SLOTS = 16;  // slots per lane when a wave ranks a large bin (bins up to 1024 slots)
constexpr int CSR_LDS_BYTES = 144 * 1024;

// exclusive scan of cnt[0..n) in place (one workgroup); afterwards cnt[k] = #slots in bins < k
template <typename CntPtr>
__device__ void block_exclusive_scan(CntPtr cnt, int n, int *s_wave /* CSR_BLOCK/64 + 1 ints */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + CSR_BLOCK - 1) / CSR_BLOCK;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += cnt[k];
    int incl = sum;  // inclusive scan of the per-thread sums across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < CSR_BLOCK / 64; ++w) {
            int v = s_wave[w];
            s_wave[w] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = s_wave[wave] + incl - sum;
    for (int k = lo; k < hi; ++k) {
        int v = cnt[k];
        cnt[k] = run;
        run += v;
    }
    __syncthreads();
}


// OrdT: uint16_t when L <= 65536 (LDS variant), int otherwise.
template <typename OrdT>
__device__ __forceinline__ void wave_sort_any(OrdT *bin, int n, int lane)
{
    if (n <= 64) wave_sort_bin<1>(bin, n, lane);
    else if (n <= 128) wave_sort_bin<2>(bin, n, lane);
    else if (n <= 256) wave_sort_bin<4>(bin, n, lane);
    else if (n <= 512) wave_sort_bin<8>(bin, n, lane);
    else wave_sort_bin<16>(bin, n, lane);
}

template <typename OrdT, bool IN_LDS>
__global__ __launch_bounds__(CSR_BLOCK) void csr_transpose_kernel(const int64_t *__restrict__ idx, int L, int nbins,
                                                                   int div, const float *__restrict__ weight,
                                                                   int *__restrict__ start, int *__restrict__ order,
                                                                   float *__restrict__ wsorted,
                                                                   int *__restrict__ scratch_ord)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_wave[32];  // 128 B: keeps the dynamic region 16-byte aligned
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t *ib = idx + (size_t)b * L;
    int *g_start = start + (size_t)b * (nbins + 1);
    int *g_order = order + (size_t)b * L;
    // gridDim.y workgroups share a cloud (in-LDS tables only): each builds the full histogram (cheap) but fills, sorts
    // and writes only its contiguous range of bins -- the per-bin sorts are what the pass spends its time on
    const int part = blockIdx.y, parts = gridDim.y;
    const int k_lo = (int)((int64_t)nbins * part / parts), k_hi = (int)((int64_t)nbins * (part + 1) / parts);

    int *cnt;   // nbins ints: histogram -> bin start -> bin end
    OrdT *ord;  // L slot ids
    if (IN_LDS) {
        cnt = reinterpret_cast<int *>(smem);
        ord = reinterpret_cast<OrdT *>(smem + (((size_t)nbins * 4 + 15) & ~(size_t)15));
    } else {
        cnt = g_start;  // reuse the output array (entry nbins is written at the end)
        ord = reinterpret_cast<OrdT *>(scratch_ord + (size_t)b * L);
    }

    for (int k = tid; k < nbins; k += CSR_BLOCK) cnt[k] = 0;
    __syncthreads();
    // Both slot passes were chains of (global index load -> LDS atomic), one 1024-slot window per iteration: ~2 us of
    // L2 latency each, 2 x L/1024 times, with one workgroup per cloud.  The indices of PER windows are now fetched
    // together into registers (and kept for the second pass when the cloud has at most PER windows), so the windows
    // themselves only touch LDS.  The fill pass still walks the windows in order with a barrier between them: a bin
    // must receive its slots in (nearly) ascending order or the insertion sort below degenerates (the padded tail of
    // a dense ball query repeats one index up to nsample times).
    constexpr int PER = 32;
    int vals[PER];
    const bool keep = L <= PER * CSR_BLOCK;
    for (int base = 0; base < L; base += PER * CSR_BLOCK) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int s = base + u * CSR_BLOCK + tid;
            vals[u] = s < L ? min(max((int)ib[s], 0), nbins - 1) : -1;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (vals[u] >= 0) atomicAdd(&cnt[vals[u]], 1);
    }
    __syncthreads();
    block_exclusive_scan(cnt, nbins, s_wave);
    const int j_lo = k_lo < nbins ? cnt[k_lo] : L;  // first slot position of this workgroup's bins (read before the fill)
    __syncthreads();
    for (int base = 0; base < L; base += PER * CSR_BLOCK) {
        if (!keep) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int s = base + u * CSR_BLOCK + tid;
                vals[u] = s < L ? min(max((int)ib[s], 0), nbins - 1) : -1;
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (base + u * CSR_BLOCK < L) {  // workgroup-uniform: the barrier is reached by every thread
                if (vals[u] >= k_lo && vals[u] < k_hi) {
                    const int pos = atomicAdd(&cnt[vals[u]], 1);  // cnt[k] ends as the END of bin k (own bins)
                    ord[pos] = (OrdT)(base + u * CSR_BLOCK + tid);
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    // canonical order inside every bin: ascending slot id.  Small bins: one thread each, insertion sort (arrival order
    // is nearly sorted).  Large bins -- a point referenced by many slots: the padded tail of a dense ball query repeats
    // its first hit up to nsample times per query, an interpolation table references each coarse point ~100 times --
    // would serialise hundreds of dependent LDS steps in one thread (measured: 78 of 95 us for SA2's table), so a whole
    // wave sorts such a bin instead: its slots sit in registers (up to 16 per lane) and go through a bitonic network
    // (in-lane exchanges for strides >= 64, wave shuffles below).
    for (int k = k_lo + tid; k < k_hi; k += CSR_BLOCK) {
        const int lo = k == k_lo ? j_lo : cnt[k - 1], hi = cnt[k];
        if (hi - lo > CSR_SMALL_BIN) continue;  // sorted by a wave below
        for (int a = lo + 1; a < hi; ++a) {
            OrdT v = ord[a];
            int p = a;
            while (p > lo && ord[p - 1] > v) {
                ord[p] = ord[p - 1];
                --p;
            }
            ord[p] = v;
        }
    }
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int k0 = k_lo + wave * 64; k0 < k_hi; k0 += (CSR_BLOCK / 64) * 64) {
            const int k = k0 + lane;
            int lo = 0, hi = 0;
            if (k < k_hi) {
                lo = k == k_lo ? j_lo : cnt[k - 1];
                hi = cnt[k];
            }
            unsigned long long big = __ballot(hi - lo > CSR_SMALL_BIN);
            while (big) {
                const int l = __builtin_ctzll(big);
                big &= big - 1;
                const int blo = __builtin_amdgcn_readlane(lo, l), bhi = __builtin_amdgcn_readlane(hi, l);
                if (bhi - blo <= 64 * CSR_RANK_SLOTS) {
                    wave_sort_any(ord + blo, bhi - blo, lane);
                    continue;
                }
                // A bin of more than 1024 slots (the first hit of many padded dense-ball queries): the fill pass above
                // walked the slots one 1024-slot window at a time with a barrier in between, so the bin is a sequence
                // of per-window segments that are already in window order -- sorting each segment (<= 1024 slots, found
                // by bisection on slot / 1024) sorts the bin.  One thread's insertion sort took 2.5 ms here.
                int a = blo;
                while (a < bhi) {
                    const int w = (int)ord[a] / CSR_BLOCK;
                    int x = a + 1, y = bhi;  // first position in (a, bhi] whose window differs
                    while (x < y) {
                        const int mid = (x + y) >> 1;
                        if ((int)ord[mid] / CSR_BLOCK == w) x = mid + 1;
                        else y = mid;
                    }
                    if (x - a > 1) wave_sort_any(ord + a, x - a, lane);
                    a = x;
                }
            }
        }
    }
    __syncthreads();
    const int j_hi = k_hi > k_lo ? cnt[k_hi - 1] : j_lo;
    for (int j = j_lo + tid; j < j_hi; j += CSR_BLOCK) {
        const int s = (int)ord[j];
        if (wsorted) wsorted[(size_t)b * L + j] = weight[(size_t)b * L + s];
        g_order[j] = s / div;
    }
    if (IN_LDS) {
        for (int k = k_lo + tid; k < k_hi; k += CSR_BLOCK) g_start[k] = k == k_lo ? j_lo : cnt[k - 1];
        if (tid == 0 && part == parts - 1) g_start[nbins] = L;
    } else {
        // cnt aliases g_start and holds bin ENDS: shift by one bin (every thread reads before anyone writes)
        const int per = (nbins + CSR_BLOCK - 1) / CSR_BLOCK;
        const int lo = min(tid * per, nbins), hi = min(lo + per, nbins);
        int prev = lo ? cnt[lo - 1] : 0;
        __syncthreads();
        for (int k = lo; k < hi; ++k) {
            int e = cnt[k];
            g_start[k] = prev;
            prev = e;
        }
        if (tid == 0) g_start[nbins] = L;
    }
}

bool csr_fits_lds(int L, int nbins);
size_t csr_lds_bytes(int L, int nbins) { return (((size_t)nbins * 4 + 15) & ~(size_t)15) + (size_t)L * 2; }

bool csr_fits_lds(int L, int nbins) { return L <= 65536 && csr_lds_bytes(L, nbins) <= (size_t)CSR_LDS_BYTES; }

// Enqueue the transpose. start: B*(nbins+1) ints, order: B*L ints, wsorted: B*L floats or null,
// scratch_ord: B*L ints, only touched when the tables do not fit LDS.
int csr_transpose(const int64_t *idx, int B, int L, int nbins, int div, const float *weight, int *start, int *order,
                  float *wsorted, int *scratch_ord, hipStream_t s)
{
    const size_t lds = csr_lds_bytes(L, nbins);
    if (L <= 65536 && lds <= (size_t)CSR_LDS_BYTES) {
        static bool attr_set[64] = {false};
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&csr_transpose_kernel<uint16_t, true>), CSR_LDS_BYTES,
                                attr_set);
        // few clouds with large tables leave most of the chip idle: up to four workgroups per cloud, each a bin range
        int parts = 1;
        while (parts < 4 && B * parts * 2 <= 256 && nbins >= parts * 2 * 64 && L >= 8192) parts *= 2;
        hipLaunchKernelGGL((csr_transpose_kernel<uint16_t, true>), dim3(B, parts), dim3(CSR_BLOCK), lds, s, idx, L, nbins,
                           div, weight, start, order, wsorted, scratch_ord);
    } else {
        hipLaunchKernelGGL((csr_transpose_kernel<int, false>), dim3(B), dim3(CSR_BLOCK), 0, s, idx, L, nbins, div,
                           weight, start, order, wsorted, scratch_ord);
    }
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Gather-sum over the transposed table: out[b,c,k] = sum_{j in [start[k], start[k+1])} w[j] * rows[b,c,order[j]]
// CC channel rows are staged in LDS (coalesced read of grad_out, each byte fetched once); every lane owns a
// destination k, reads its (order, weight) run once and accumulates CC channels from LDS.
constexpr int GS_BLOCK = 512;
constexpr int GS_HUB_MIN = 128;   // longer runs are summed by the whole workgroup
constexpr int GS_HUB_CAP = 1024;  // hubs listed per (cloud, channel group); further ones are walked by their lane

template <int CC, bool WEIGHTED, bool IN_LDS>
__global__ __launch_bounds__(GS_BLOCK) void gather_sum_kernel(const float *__restrict__ rows,
                                                               const int *__restrict__ start,
                                                               const int *__restrict__ order,
                                                               const float *__restrict__ wsorted, int C, int nbins,
                                                               int Lrow, int Lslots, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *srow = reinterpret_cast<float *>(smem);
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * CC;
    const int tid = threadIdx.x;
    const float *gbase = rows + ((size_t)b * C + c0) * Lrow;
    const int nc = min(CC, C - c0);
    if (IN_LDS) {
        const int total = nc * Lrow;
        if ((Lrow & 3) == 0) {
            const float4 *g4 = reinterpret_cast<const float4 *>(gbase);
            float4 *s4 = reinterpret_cast<float4 *>(srow);
            for (int e = tid; e < total / 4; e += GS_BLOCK) s4[e] = g4[e];
        } else {
            for (int e = tid; e < total; e += GS_BLOCK) srow[e] = gbase[e];
        }
        __syncthreads();
    }
    const float *src = IN_LDS ? srow : gbase;
    const int *st = start + (size_t)b * (nbins + 1);
    const int *od = order + (size_t)b * Lslots;
    const float *ws = WEIGHTED ? wsorted + (size_t)b * Lslots : nullptr;
    // Destinations with more than GS_HUB_MIN slots (the shared first hit of padded ball queries collects hundreds to
    // thousands) are not walked by one lane while its workgroup waits: they are listed and then summed by the whole
    // workgroup, every wave a contiguous piece of the run, every lane a stride of the piece, the partial sums added in a
    // fixed order (lanes by butterfly, then pieces in order): reproducible run to run; shorter runs keep the oracle's
    // sequential order bit for bit.  (128 centres x 128 slots over 512 points, 128 channels: 1.5 ms -> see DESIGN.md.)
    __shared__ int s_nhub;
    __shared__ int s_hub[GS_HUB_CAP];
    __shared__ float s_part[GS_BLOCK / 64][CC];
    if (tid == 0) s_nhub = 0;
    __syncthreads();
    for (int k = tid; k < nbins; k += GS_BLOCK) {
        const int lo = st[k], hi = st[k + 1];
        if (hi - lo > GS_HUB_MIN) {
            const int pos = atomicAdd(&s_nhub, 1);
            if (pos < GS_HUB_CAP) {
                s_hub[pos] = k;
                continue;
            }  // (list full: this lane walks it after all)
        }
        float acc[CC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) acc[cc] = 0.0f;
        for (int j = lo; j < hi; ++j) {
            const int r = od[j];
            const float w = WEIGHTED ? ws[j] : 1.0f;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                if (cc < nc) {
                    const float v = src[(size_t)cc * Lrow + r];
                    acc[cc] = acc[cc] + (WEIGHTED ? w * v : v);  // mul then add, never fused (oracle order)
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
            if (cc < nc) out[((size_t)b * C + c0 + cc) * nbins + k] = acc[cc];
    }
    __syncthreads();
    const int nhub = min(s_nhub, GS_HUB_CAP);
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = GS_BLOCK / 64;
    for (int h = 0; h < nhub; ++h) {
        const int k = s_hub[h];
        const int lo = st[k], hi = st[k + 1];
        const int per = (hi - lo + NW - 1) / NW;
        const int plo = min(lo + wave * per, hi), phi = min(plo + per, hi);
        float acc[CC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) acc[cc] = 0.0f;
        for (int j = plo + lane; j < phi; j += 64) {
            const int r = od[j];
            const float w = WEIGHTED ? ws[j] : 1.0f;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nc) {
                    const float v = src[(size_t)cc * Lrow + r];
                    acc[cc] = acc[cc] + (WEIGHTED ? w * v : v);
                }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[cc] = acc[cc] + __shfl_xor(acc[cc], off);
        if (lane == 0)
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) s_part[wave][cc] = acc[cc];
        __syncthreads();
        if (tid < nc) {
            float sum = s_part[0][tid];
            for (int w = 1; w < NW; ++w) sum = sum + s_part[w][tid];
            out[((size_t)b * C + c0 + tid) * nbins + k] = sum;
        }
        __syncthreads();
    }
}

template <int CC, bool WEIGHTED, bool IN_LDS>
static void launch_gather(const float *rows, const int *start, const int *order, const float *wsorted, int B, int C,
                          int nbins, int Lrow, int Lslots, float *out, hipStream_t s)
{
    const size_t lds = IN_LDS ? (size_t)CC * Lrow * sizeof(float) : 0;
    if (lds > 64 * 1024) {
        static bool attr_set[64] = {false};
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&gather_sum_kernel<CC, WEIGHTED, IN_LDS>), CSR_LDS_BYTES,
                                attr_set);
    }
    dim3 grid((C + CC - 1) / CC, B);
    hipLaunchKernelGGL((gather_sum_kernel<CC, WEIGHTED, IN_LDS>), grid, dim3(GS_BLOCK), lds, s, rows, start, order,
                       wsorted, C, nbins, Lrow, Lslots, out);
}

template <bool WEIGHTED>
static int gather_sum_t(const float *rows, const int *start, const int *order, const float *wsorted, int B, int C,
                        int nbins, int Lrow, int Lslots, float *out, hipStream_t s)
{
    const size_t row_bytes = (size_t)Lrow * sizeof(float);
    if (4 * row_bytes <= (size_t)CSR_LDS_BYTES && C >= 4)
        launch_gather<4, WEIGHTED, true>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s);
    else if (2 * row_bytes <= (size_t)CSR_LDS_BYTES && C >= 2)
        launch_gather<2, WEIGHTED, true>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s);
    else if (row_bytes <= (size_t)CSR_LDS_BYTES)
        launch_gather<1, WEIGHTED, true>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s);
    else
        launch_gather<4, WEIGHTED, false>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s);
    return check_launch();
}

int gather_sum(const float *rows, const int *start, const int *order, const float *wsorted, int B, int C, int nbins,
               int Lrow, int Lslots, float *out, hipStream_t s)
{
    return wsorted ? gather_sum_t<true>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s)
                   : gather_sum_t<false>(rows, start, order, wsorted, B, C, nbins, Lrow, Lslots, out, s);
}

// Workspace carve shared by the two backward entry points (all offsets 16-byte aligned).
ScatterWorkspace carve_scatter_workspace(void *ws, int B, int L, int nbins, bool with_weights)
{
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    ScatterWorkspace w;
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    w.start = reinterpret_cast<int *>(p + off);
    off += up((size_t)B * (nbins + 1) * 4);
    w.order = reinterpret_cast<int *>(p + off);
    off += up((size_t)B * L * 4);
    w.scratch = reinterpret_cast<int *>(p + off);
    off += up((size_t)B * L * 4);
    w.wsorted = nullptr;
    if (with_weights) {
        w.wsorted = reinterpret_cast<float *>(p + off);
        off += up((size_t)B * L * 4);
    }
    w.merge_tmp = reinterpret_cast<int *>(p + off);
    off += up((size_t)B * L * 4);
    w.hubs = reinterpret_cast<int *>(p + off);
    off += up(((size_t)B * nbins + 1) * 4);
    w.bytes = off;
    return w;
}

}  // namespace tp3d

TP3D_EXPORT size_t tp3d_scatter_workspace_bytes(int B, int L, int nbins, int with_weights)
{
    if (B < 0 || L < 0 || nbins < 0) return 0;
    return tp3d::carve_scatter_workspace(nullptr, B, L, nbins, with_weights != 0).bytes;
}
