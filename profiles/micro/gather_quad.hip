// Micro-benchmark behind the traversal kernels' fetch: every lane chases its own chain of random 64-byte records of a table
// that lives in L2 / Infinity Cache (the BVH: 6 MB), 7 waves per SIMD.  Variant A: four 16-byte loads per lane and step (64 lanes
// x 4 instructions = 256 line requests per wave step).  Variant B: the four lanes of a quad fetch the quad's four records
// together -- instruction k loads record k, lane j its j-th 16 bytes (16 lines per instruction) -- and transpose with DPP.
//   hipcc -O3 --offload-arch=gfx950 -o gather_quad gather_quad.hip && ./gather_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ uint32_t quadBcast(uint32_t v, int k)
{
    switch (k) {
    case 0: return uint32_t(__builtin_amdgcn_mov_dpp(int(v), 0x00, 0xF, 0xF, true));
    case 1: return uint32_t(__builtin_amdgcn_mov_dpp(int(v), 0x55, 0xF, 0xF, true));
    case 2: return uint32_t(__builtin_amdgcn_mov_dpp(int(v), 0xAA, 0xF, 0xF, true));
    default: return uint32_t(__builtin_amdgcn_mov_dpp(int(v), 0xFF, 0xF, 0xF, true));
    }
}
__device__ __forceinline__ float swapXor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); } // quad_perm [1,0,3,2]
__device__ __forceinline__ float swapXor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); } // quad_perm [2,3,0,1]

// exchange so that lane j ends with piece i of ITS record in r[i]  (before: r[k] = piece j of record k of the quad)
__device__ __forceinline__ void transpose4(float4 r[4], int j)
{
    const bool odd = j & 1, hi = j & 2;
#define TG_X1(a, b, c) { float s = odd ? r[a].c : r[b].c; float g = swapXor1(s); if (odd) r[a].c = g; else r[b].c = g; }
    TG_X1(0, 1, x) TG_X1(0, 1, y) TG_X1(0, 1, z) TG_X1(0, 1, w) TG_X1(2, 3, x) TG_X1(2, 3, y) TG_X1(2, 3, z) TG_X1(2, 3, w)
#undef TG_X1
#define TG_X2(a, b, c) { float s = hi ? r[a].c : r[b].c; float g = swapXor2(s); if (hi) r[a].c = g; else r[b].c = g; }
    TG_X2(0, 2, x) TG_X2(0, 2, y) TG_X2(0, 2, z) TG_X2(0, 2, w) TG_X2(1, 3, x) TG_X2(1, 3, y) TG_X2(1, 3, z) TG_X2(1, 3, w)
#undef TG_X2
}

template <int MODE>
__global__ void __launch_bounds__(256, 7) chase(const float4 *table, uint32_t mask, int steps, uint32_t *out)
{
    const uint32_t tid = blockIdx.x*blockDim.x + threadIdx.x;
    uint32_t idx = tid*2654435761u & mask;
    const int j = threadIdx.x & 3;
    float acc = 0.0f;
    for (int s = 0; s < steps; ++s) {
        float4 r[4];
        if (MODE == 0) {
            const float4 *p = table + size_t(idx)*4;
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = table[size_t(quadBcast(idx, k))*4 + j];
            transpose4(r, j);
        }
        acc += r[0].x + r[1].y + r[2].z + r[3].w;
        idx = ((__float_as_uint(r[0].x) ^ __float_as_uint(r[3].w)*0x9E3779B9u) + tid*2654435761u + uint32_t(s)*40503u) & mask; // next record: from both ends of this one, per-lane walk
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = __float_as_uint(acc) ^ idx;
}

int main()
{
    const uint32_t nRec = 1u << 17; // 128 Ki records x 64 B = 8 MB
    std::vector<float> h(size_t(nRec)*16);
    uint32_t x = 12345;
    for (auto &f : h) { x = x*1664525u + 1013904223u; uint32_t u = x; memcpy(&f, &u, 4); }
    float4 *d; uint32_t *out;
    const int blocks = 2048, steps = 400;
    hipMalloc(&d, h.size()*4); hipMemcpy(d, h.data(), h.size()*4, hipMemcpyHostToDevice);
    hipMalloc(&out, blocks*256*4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> ra(blocks*256), rb(blocks*256);
    for (int rep = 0; rep < 3; ++rep) {
        float ms[2];
        for (int m = 0; m < 2; ++m) {
            hipEventRecord(e0);
            if (m == 0) chase<0><<<blocks, 256>>>(d, nRec - 1, steps, out); else chase<1><<<blocks, 256>>>(d, nRec - 1, steps, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[m], e0, e1);
            hipMemcpy(m ? rb.data() : ra.data(), out, blocks*256*4, hipMemcpyDeviceToHost);
        }
        const double rec = double(blocks)*256*steps;
        printf("per-lane 4x16B: %.3f ms (%.1f G records/s, %.2f TB/s)   quad-cooperative + DPP transpose: %.3f ms (%.1f G records/s, %.2f TB/s)   same results: %s\n",
               ms[0], rec/ms[0]/1e6, rec*64/ms[0]/1e9, ms[1], rec/ms[1]/1e6, rec*64/ms[1]/1e9, ra == rb ? "yes" : "NO");
    }
    return 0;
}
