// Which engine does a D2H hipMemcpyAsync use on this stack?  (diagnostic; see profiles/r01_experiments.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void fill(int *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int)i;
}
__global__ void spin(float *p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
int main() {
    const size_t n = 25u << 20;  // 100 MB
    int *dev, *host;
    float *scratch;
    hipMalloc(&dev, n * 4);
    hipMalloc(&scratch, 4096);
    hipHostMalloc(&host, n * 4, hipHostMallocDefault);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int rep = 0; rep < 3; ++rep) {
        fill<<<1024, 256, 0, a>>>(dev, n);
        hipEventRecord(ev, a);
        hipStreamWaitEvent(b, ev, 0);
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, b);
        spin<<<256, 256, 0, a>>>(scratch, 1000);
        hipStreamSynchronize(b);
        auto t1 = std::chrono::steady_clock::now();
        printf("rep %d: D2H %.3f ms  host[12345]=%d\n", rep, std::chrono::duration<double, std::milli>(t1 - t0).count(), host[12345]);
        hipDeviceSynchronize();
    }
    return 0;
}
