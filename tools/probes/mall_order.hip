// Does the traversal direction of a consumer matter for the 256 MiB Infinity Cache?  A producer kernel writes a tensor
// front to back; a consumer (out = 2 * in, float4 grid-stride) then reads it front to back or back to front.  With an
// LRU-like memory-side cache the back of the tensor is still resident when the consumer starts, so the reverse walk should
// hit for the first part of its reads.      hipcc --offload-arch=gfx950 -O3 tools/probes/mall_order.hip -o /tmp/mall_order
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void produce(float4 *t, size_t n4, float v)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) t[e] = make_float4(v, v, v, v);
}
template <bool REV>
__global__ __launch_bounds__(256) void consume(const float4 *in, float4 *out, size_t n4)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const size_t i = REV ? n4 - 1 - e : e;
        float4 v = in[i];
        out[i] = make_float4(v.x * 2, v.y * 2, v.z * 2, v.w * 2);
    }
}
template <bool REV>
__global__ __launch_bounds__(256) void reduce_only(const float4 *in, float *out, size_t n4)
{
    float s = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const float4 v = in[REV ? n4 - 1 - e : e];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float *scratch;
    hipMalloc(&scratch, 1024);
    for (size_t mb : {64, 134, 268, 537}) {
        const size_t n4 = mb * 1000000 / 16;
        float4 *a, *b;
        hipMalloc(&a, n4 * 16);
        hipMalloc(&b, n4 * 16);
        const int grid = 2048;
        for (int rev = 0; rev < 2; ++rev)
            for (int kind = 0; kind < 2; ++kind) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    produce<<<grid, 256>>>(a, n4, 1.0f + rep);
                    hipEventRecord(e0);
                    if (kind == 0) {
                        if (rev) consume<true><<<grid, 256>>>(a, b, n4); else consume<false><<<grid, 256>>>(a, b, n4);
                    } else {
                        if (rev) reduce_only<true><<<grid, 256>>>(a, scratch, n4); else reduce_only<false><<<grid, 256>>>(a, scratch, n4);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double bytes = (kind == 0 ? 2.0 : 1.0) * n4 * 16;
                printf("%4zu MB tensor, %s, %s walk: %7.1f us  %6.2f TB/s\n", mb, kind == 0 ? "read+write" : "read only ",
                       rev ? "reverse" : "forward", best * 1e3, bytes / best / 1e9);
            }
        hipFree(a);
        hipFree(b);
    }
    return 0;
}
