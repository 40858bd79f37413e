// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this repo's kernels (MI355X_MICROARCH.md "HBM": only wide
// coalesced streaming reads are calibrated there -- FETCH_SIZE reports half their bytes -- "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three kernels over an 8 GiB table (far beyond the 256 MiB Infinity Cache,
// so every access misses it), each touching a KNOWN number of distinct 128-byte lines exactly once:
//   stream16   : coalesced 16 B per lane, the whole table                                   (the guide's calibrated case)
//   record64   : every lane reads one random, distinct 64-byte record as 4 x 16 B            (BVH nodes / triangles / path records)
//   word4      : every lane reads one random, distinct 4-byte word, one per 128-byte line    (state words, list entries after sorting)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and divide the counter (KiB) by the printed access counts; the program itself
// prints the time per kernel, i.e. the rate the memory system sustains for each pattern.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// a bijection on [0, 2^bits): odd multiplier + xorshift, so that every index is visited exactly once in a scattered order
__device__ __forceinline__ uint64_t scatter(uint64_t i, int bits)
{
    const uint64_t mask = (1ull << bits) - 1ull;
    i = (i*0x9E3779B97F4A7C15ull) & mask;
    i ^= i >> (bits/2);
    i = (i*0xD6E8FEB86659FD93ull) & mask;
    return i;
}

__global__ void stream16(const float4 *t, size_t n, float *sink)
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x*size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x)*blockDim.x) {
        float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void record64(const float4 *t, int bits, size_t count, float *sink) // table = 2^bits records of 64 B
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x*size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x)*blockDim.x) {
        const float4 *r = t + scatter(i, bits)*4;
        float4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void word4(const float *t, int bits, size_t count, float *sink) // table = 2^bits lines of 128 B, one word of each
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x*size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x)*blockDim.x)
        acc += t[scatter(i, bits)*32 + (i & 31)];
    if (acc == 123.456f) *sink = acc;
}

int main()
{
    const size_t bytes = 8ull << 30;
    float *table, *sink;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256*8, block = 256;
    const size_t nVec = bytes/16, nRec = size_t(1) << 26 /* of 2^27 records */, nWord = size_t(1) << 25 /* of 2^26 lines */;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        CHECK(hipEventRecord(e0));
        stream16<<<grid, block>>>(reinterpret_cast<const float4 *>(table), nVec, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("stream16: %zu bytes in %.3f ms = %.2f TB/s\n", bytes, ms, bytes/ms/1e9);
        CHECK(hipEventRecord(e0));
        record64<<<grid, block>>>(reinterpret_cast<const float4 *>(table), 27, nRec, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("record64: %zu records (64 B each, distinct) in %.3f ms = %.2f G records/s = %.2f TB/s of records\n", nRec, ms, nRec/ms/1e6, nRec*64/ms/1e9);
        CHECK(hipEventRecord(e0));
        word4<<<grid, block>>>(table, 26, nWord, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("word4:    %zu words (one per 128-byte line, distinct) in %.3f ms = %.2f G words/s\n", nWord, ms, nWord/ms/1e6);
    }
    return 0;
}
