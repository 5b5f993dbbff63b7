// Dependent-load (pointer-chase) latency on the GPU box: one 64-lane wave per workgroup, each lane chases
// its own random cycle through a buffer of `mb` MiB; W workgroups run concurrently.  Prints ns per
// dependent load.  This is the round-trip cost a BCP step pays ~5 times.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chase scripts/microbench/chase.hip && /tmp/chase
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

__global__ void chase(const uint32_t* __restrict__ next, uint32_t n, uint32_t steps, uint32_t* out) {
    // every lane starts somewhere else on the one big cycle (entries sit at a 64-byte stride: index = line * 16)
    uint32_t i = (uint32_t)(((uint64_t)(blockIdx.x * 64 + threadIdx.x) * 2654435761u) % (n / 16)) * 16;
    for (uint32_t s = 0; s < steps; s++) i = next[i];
    if (i == 0xffffffffu) out[0] = i;   // keep the chain alive
}

// chase MB W STEPS: ONE configuration, for calibrating rocprofv3's FETCH_SIZE on this access pattern (every dependent
// load touches one random 64-byte line: W * 64 * STEPS lines in the timed launch, printed as "expected bytes").
int main(int argc, char** argv) {
    std::mt19937_64 rng(1);
    const bool one = argc == 4;
    std::vector<size_t> sizes = one ? std::vector<size_t>{(size_t)atol(argv[1])} : std::vector<size_t>{8, 256, 8192, 65536};
    for (size_t mb : sizes) {
        const size_t n = mb * 1024 * 1024 / 64;          // one entry per 64-byte line
        std::vector<uint32_t> perm(n), next(n * 16, 0);
        std::iota(perm.begin(), perm.end(), 0u);
        for (size_t i = n - 1; i > 0; i--) std::swap(perm[i], perm[rng() % (i + 1)]);
        for (size_t i = 0; i < n; i++) next[(size_t)perm[i] * 16] = (uint32_t)(perm[(i + 1) % n] * 16);   // one cycle, 64-byte stride
        uint32_t *d_next, *d_out;
        if (hipMalloc(&d_next, next.size() * 4) != hipSuccess) { printf("alloc %zu MiB failed\n", mb); continue; }
        (void)hipMalloc(&d_out, 4);
        (void)hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice);
        std::vector<int> ws = one ? std::vector<int>{atoi(argv[2])} : std::vector<int>{1, 256, 1024, 4096, 8192};
        for (int W : ws) {
            const uint32_t steps = one ? (uint32_t)atol(argv[3]) : (W >= 1024 ? 2000 : 20000);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(chase, dim3(W), dim3(64), 0, 0, d_next, (uint32_t)(n * 16), 100u, d_out);   // warm-up
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(chase, dim3(W), dim3(64), 0, 0, d_next, (uint32_t)(n * 16), steps, d_out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("buffer %6zu MiB, %4d waves x 64 lanes (64 independent chains per wave): %8.1f ns per dependent load, %7.1f GB/s of 64-byte lines\n",
                   mb, W, ms * 1e6 / steps, (double)W * 64 * steps * 64 / (ms * 1e-3) / 1e9);
            if (one) printf("expected bytes of the timed launch (64-byte lines): %.0f\n", (double)W * 64 * steps * 64);
        }
        (void)hipFree(d_next); (void)hipFree(d_out);
    }
    return 0;
}
