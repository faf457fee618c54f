// Micro-benchmark (MI355X): cost of scattered 320-byte row traffic by access shape.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/rowio.hip -o /tmp/rowio && /tmp/rowio
// rows: 200,000 x 80 floats; a launch touches `n` distinct random rows, one 16-lane quarter-wave (dword shapes) or 20 lanes
// (dwordx4 shapes) per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int STRIDE = 80;

template <int MODE>  // 0 load dword, 1 store dword, 2 atomic dword, 3 load+store dword (rmw)
__global__ __launch_bounds__(256) void k_dword(float* tab, const int* ids, int n, float* sink) {
  const int sub = (blockIdx.x * 256 + threadIdx.x) >> 4, j = threadIdx.x & 15;
  if (sub >= n) return;
  float* p = tab + (long)ids[sub] * STRIDE + j;
  float acc = 0.f;
  if (MODE == 0 || MODE == 3) {
#pragma unroll
    for (int k = 0; k < 5; ++k) acc += p[k * 16];
  }
  if (MODE == 1 || MODE == 3) {
#pragma unroll
    for (int k = 0; k < 5; ++k) p[k * 16] = acc + (float)k;
  }
  if (MODE == 2) {
#pragma unroll
    for (int k = 0; k < 5; ++k) __hip_atomic_fetch_add(p + k * 16, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 0 && acc == 1.2345f) sink[0] = acc;
}

template <int SCOPE>
__global__ __launch_bounds__(256) void k_atomic_scope(float* tab, const int* ids, int n) {
  const int sub = (blockIdx.x * 256 + threadIdx.x) >> 4, j = threadIdx.x & 15;
  if (sub >= n) return;
  float* p = tab + (long)ids[sub] * STRIDE + j;
#pragma unroll
  for (int k = 0; k < 5; ++k) __hip_atomic_fetch_add(p + k * 16, 1.0f, __ATOMIC_RELAXED, SCOPE);
}
// device-scope (write-through / L2-bypassing) plain accesses: what a cross-wavefront hand-over of a row would use
template <int MODE>  // 0 load, 1 store, 3 both
__global__ __launch_bounds__(256) void k_dword_sc(float* tab, const int* ids, int n, float* sink) {
  const int sub = (blockIdx.x * 256 + threadIdx.x) >> 4, j = threadIdx.x & 15;
  if (sub >= n) return;
  float* p = tab + (long)ids[sub] * STRIDE + j;
  float acc = 0.f;
  if (MODE == 0 || MODE == 3) {
#pragma unroll
    for (int k = 0; k < 5; ++k) acc += __hip_atomic_load(p + k * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 1 || MODE == 3) {
#pragma unroll
    for (int k = 0; k < 5; ++k) __hip_atomic_store(p + k * 16, acc + (float)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 0 && acc == 1.2345f) sink[0] = acc;
}

template <int MODE>  // 0 load x4, 1 store x4, 3 rmw x4: chunk c of the launch -> row c / 20, piece c % 20
__global__ __launch_bounds__(256) void k_x4(float* tab, const int* ids, int n, float* sink) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(c / 20), off = (int)(c - (long)row * 20);
  if (row >= n) return;
  float4* p = (float4*)(tab + (long)ids[row] * STRIDE) + off;
  float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  if (MODE == 0 || MODE == 3) v = *p;
  if (MODE == 1 || MODE == 3) { v.x += 1.f; *p = v; }
  if (MODE == 0 && v.x == 1.2345f) sink[0] = v.x;
}

int main() {
  const int NE = 200000, N = 65536;
  float *tab, *sink; int* ids;
  CK(hipMalloc(&tab, (size_t)NE * STRIDE * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ids, N * 4));
  CK(hipMemset(tab, 0, (size_t)NE * STRIDE * 4));
  std::vector<int> perm(NE); std::iota(perm.begin(), perm.end(), 0);
  std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
  CK(hipMemcpy(ids, perm.data(), N * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < 30; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); tot += ms; }
    printf("%-22s avg %7.2f us  min %7.2f us   %6.2f TB/s (min)\n", name, tot / 30 * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e12);
    return 0;
  };
  const double rb = (double)N * 320;
  const int gd = (N * 16 + 255) / 256, gx = (int)(((long)N * 20 + 255) / 256);
  run("load  dword  (16/row)", [&] { hipLaunchKernelGGL(k_dword<0>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("load  dwordx4 (20/row)", [&] { hipLaunchKernelGGL(k_x4<0>, dim3(gx), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("store dword", [&] { hipLaunchKernelGGL(k_dword<1>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("store dwordx4", [&] { hipLaunchKernelGGL(k_x4<1>, dim3(gx), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("atomic dword", [&] { hipLaunchKernelGGL(k_dword<2>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("atomic dword wg-scope", [&] { hipLaunchKernelGGL(k_atomic_scope<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(gd), dim3(256), 0, 0, tab, ids, N); }, rb);
  run("atomic dword wave-scope", [&] { hipLaunchKernelGGL(k_atomic_scope<__HIP_MEMORY_SCOPE_WAVEFRONT>, dim3(gd), dim3(256), 0, 0, tab, ids, N); }, rb);
  run("atomic dword sys-scope", [&] { hipLaunchKernelGGL(k_atomic_scope<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(gd), dim3(256), 0, 0, tab, ids, N); }, rb);
  run("load  dword agent-sc", [&] { hipLaunchKernelGGL(k_dword_sc<0>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("store dword agent-sc", [&] { hipLaunchKernelGGL(k_dword_sc<1>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, rb);
  run("rmw   dword agent-sc", [&] { hipLaunchKernelGGL(k_dword_sc<3>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, 2 * rb);
  run("rmw   dword", [&] { hipLaunchKernelGGL(k_dword<3>, dim3(gd), dim3(256), 0, 0, tab, ids, N, sink); }, 2 * rb);
  run("rmw   dwordx4", [&] { hipLaunchKernelGGL(k_x4<3>, dim3(gx), dim3(256), 0, 0, tab, ids, N, sink); }, 2 * rb);
  for (int n2 : {16384, 32768}) {
    const int g2 = (n2 * 16 + 255) / 256;
    char nm[64]; snprintf(nm, 64, "atomic dword n=%d", n2);
    run(nm, [&] { hipLaunchKernelGGL(k_dword<2>, dim3(g2), dim3(256), 0, 0, tab, ids, n2, sink); }, (double)n2 * 320);
    snprintf(nm, 64, "store dword n=%d", n2);
    run(nm, [&] { hipLaunchKernelGGL(k_dword<1>, dim3(g2), dim3(256), 0, 0, tab, ids, n2, sink); }, (double)n2 * 320);
  }
  return 0;
}
