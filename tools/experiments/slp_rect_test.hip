// Reduction of the SLP-vectoriser finding (round 6): the rectangle test of the path tracer (rectangle.rs:38-52) as a stand-alone kernel.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/experiments/slp_rect_test.hip -o /tmp/slp_rect && /tmp/slp_rect          (SLP on: hipcc's default)
//   hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 ... && ...                                                  (SLP off)
// With the vectoriser on, the four bound comparisons become one <4 x float> fcmp + llvm.vector.reduce.and (visible with -Rpass=slp-vectorizer:
// "Vectorized horizontal reduction"). The program compares the device's verdicts with the host's on random rays against rectangles whose width
// and height differ, and prints how many differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
struct f3 { float x, y, z; };
struct Rect { float inv[12]; float width, height; };   // object-space transform rows 0..2 and the rectangle's size
__host__ __device__ inline bool rect_test(float width, float height, f3 o, f3 d, float min_t, float max_t, float& t_out) {
    if (fabsf(d.z) < 1e-8f) return false;
    float t = -o.z / d.z;
    if (t < min_t || t > max_t) return false;
    f3 p = {o.x + d.x * t, o.y + d.y * t, o.z + d.z * t};
    float hw = width / 2.0f, hh = height / 2.0f;
    if (p.x >= -hw && p.x <= hw && p.y >= -hh && p.y <= hh) { t_out = t; return true; }
    return false;
}
__host__ __device__ inline int one(const Rect* __restrict__ r, const float* __restrict__ ray, float& t) {
    const float* m = r->inv;
    f3 wo = {ray[0], ray[1], ray[2]}, wd = {ray[3], ray[4], ray[5]};
    f3 o = {m[0] * wo.x + m[1] * wo.y + m[2] * wo.z + m[3], m[4] * wo.x + m[5] * wo.y + m[6] * wo.z + m[7], m[8] * wo.x + m[9] * wo.y + m[10] * wo.z + m[11]};
    f3 d = {m[0] * wd.x + m[1] * wd.y + m[2] * wd.z, m[4] * wd.x + m[5] * wd.y + m[6] * wd.z, m[8] * wd.x + m[9] * wd.y + m[10] * wd.z};
    t = ray[7];
    return rect_test(r->width, r->height, o, d, ray[6], ray[7], t) ? 1 : 0;
}
__global__ void k(const Rect* __restrict__ rects, const float* __restrict__ rays, int n, int n_rects, int* __restrict__ hit, float* __restrict__ t_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t;
    hit[i] = one(rects + (i % n_rects), rays + 8 * (size_t)i, t);
    t_out[i] = t;
}
int main() {
    const int n = 1 << 22, n_rects = 7;
    std::vector<Rect> rects(n_rects);
    std::vector<float> rays(8 * (size_t)n);
    srand(5);
    auto u = [] { return (float)rand() / (float)RAND_MAX; };
    for (auto& r : rects) {
        for (int k = 0; k < 12; ++k) r.inv[k] = (k % 5 == 0) ? 0.5f + u() : 0.2f * (u() - 0.5f);
        r.width = 0.5f + 3.0f * u(); r.height = 0.5f + 3.0f * u();
    }
    for (size_t i = 0; i < (size_t)n; ++i) {
        float* q = &rays[8 * i];
        for (int k = 0; k < 3; ++k) { q[k] = 4.0f * (u() - 0.5f); q[3 + k] = 2.0f * (u() - 0.5f); }
        q[2] = 3.0f + u(); q[5] = -0.2f - u(); q[6] = 0.001f; q[7] = 1e30f;
    }
    Rect* d_r; float* d_rays; int* d_hit; float* d_t;
    hipMalloc(&d_r, sizeof(Rect) * n_rects); hipMalloc(&d_rays, 4 * rays.size()); hipMalloc(&d_hit, 4 * (size_t)n); hipMalloc(&d_t, 4 * (size_t)n);
    hipMemcpy(d_r, rects.data(), sizeof(Rect) * n_rects, hipMemcpyHostToDevice); hipMemcpy(d_rays, rays.data(), 4 * rays.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d_r, d_rays, n, n_rects, d_hit, d_t);
    std::vector<int> hit(n); std::vector<float> tt(n);
    hipMemcpy(hit.data(), d_hit, 4 * (size_t)n, hipMemcpyDeviceToHost); hipMemcpy(tt.data(), d_t, 4 * (size_t)n, hipMemcpyDeviceToHost);
    long bad = 0, hits = 0;
    for (int i = 0; i < n; ++i) {
        float t; const int h = one(&rects[i % n_rects], &rays[8 * (size_t)i], t);
        hits += h;
        if (h != hit[i] || (h && t != tt[i])) { if (bad < 5) printf("ray %d: host %d t %g, device %d t %g\n", i, h, t, hit[i], tt[i]); ++bad; }
    }
    printf("rectangle test, %d rays, %ld hits on the host: %ld verdicts differ between device and host\n", n, hits, bad);
    return bad ? 1 : 0;
}
