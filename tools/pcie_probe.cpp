// tools/pcie_probe.cpp -- what the host link of the GPU box gives: pinned H2D / D2H alone and together, pageable memcpy by N threads
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)768 << 20;
    void *d1, *d2; uint8_t *h1, *h2;
    hipMalloc(&d1, N); hipMalloc(&d2, N);
    hipHostMalloc((void **)&h1, N, hipHostMallocDefault); hipHostMalloc((void **)&h2, N, hipHostMallocDefault);
    memset(h1, 1, N); memset(h2, 2, N);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); double a = now() - t;
        t = now(); hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); double b = now() - t;
        t = now(); hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); double c = now() - t;
        printf("pinned H2D %.1f GB/s  D2H %.1f GB/s  both at once %.1f GB/s each (%.1f total)\n", N / a / 1e9, N / b / 1e9, N / c / 1e9, 2 * N / c / 1e9);
    }
    // slices of 8 MB, H2D back to back
    for (size_t sl : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
        double t = now();
        for (size_t o = 0; o < N; o += sl) hipMemcpyAsync((uint8_t *)d1 + o, h1 + o, sl, hipMemcpyHostToDevice, s1);
        hipStreamSynchronize(s1);
        printf("H2D in %zu MB slices: %.1f GB/s\n", sl >> 20, N / (now() - t) / 1e9);
    }
    std::vector<uint8_t> pg(N, 3), pg2(N, 4);
    for (unsigned T : {1u, 2u, 4u, 8u, 12u, 16u}) {
        double t = now();
        std::vector<std::thread> th;
        const size_t part = N / T;
        for (unsigned i = 0; i < T; i++) th.emplace_back([&, i] { memcpy(h1 + i * part, pg.data() + i * part, part); });
        for (auto &x : th) x.join();
        double a = now() - t;
        t = now();
        th.clear();
        for (unsigned i = 0; i < T; i++) th.emplace_back([&, i] { memcpy(pg2.data() + i * part, h2 + i * part, part); });
        for (auto &x : th) x.join();
        double b = now() - t;
        printf("memcpy %2u threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s\n", T, N / a / 1e9, N / b / 1e9);
    }
    // thread spawn cost
    { double t = now(); for (int i = 0; i < 200; i++) { std::thread x([] {}); x.join(); } printf("thread spawn+join: %.1f us\n", (now() - t) / 200 * 1e6); }
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    // pageable hipMemcpy directly
    { double t = now(); hipMemcpy(d1, pg.data(), N, hipMemcpyHostToDevice); printf("pageable hipMemcpy H2D %.1f GB/s\n", N / (now() - t) / 1e9);
      t = now(); hipMemcpy(pg2.data(), d2, N, hipMemcpyDeviceToHost); printf("pageable hipMemcpy D2H %.1f GB/s\n", N / (now() - t) / 1e9); }
    // register in place
    { double t = now(); hipError_t e = hipHostRegister(pg.data(), N, hipHostRegisterDefault); double r = now() - t;
      t = now(); if (e == hipSuccess) { hipMemcpy(d1, pg.data(), N, hipMemcpyHostToDevice); } double c = now() - t;
      t = now(); if (e == hipSuccess) hipHostUnregister(pg.data()); double u = now() - t;
      printf("hipHostRegister %d: register %.1f ms, copy %.1f GB/s, unregister %.1f ms\n", (int)e, r * 1e3, N / c / 1e9, u * 1e3); }
    return 0;
}
