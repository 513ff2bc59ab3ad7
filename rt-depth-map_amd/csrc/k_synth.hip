// k_synth.hip -- device twin of rt-depth-map_amd/synth.py (bit-identical, tests/test_gpu_synth.py).
// Stands in for the reference's camera + MJPEG front end (/root/reference/stream/,
// /root/reference/decoder/), which is out of scope: every rank of a multi-GPU run synthesises
// its own shard of the rectified-pair stream directly in HBM, with no communication.
#include "rtdm_kernels.h"

namespace rtdm {

struct SynthParams {
    int nrect; int rect[6][5];      // x0,y0,x1,y1,d
    int base, gx, gy, lo, hi;
    int nout; int outl[20][3];      // px,py,dalt
    int ps, fx, fy, sx, sy;
    unsigned long long seed;
};

__host__ __device__ inline unsigned long long mix(unsigned long long seed, long long a, long long b, long long c)
{
    const unsigned long long K0 = 0x9E3779B97F4A7C15ull, K1 = 0xBF58476D1CE4E5B9ull, K2 = 0x94D049BB133111EBull,
                             K3 = 0xD6E8FEB86659FD93ull, K4 = 0xA0761D6478BD642Full;
    unsigned long long z = seed * K3 + (unsigned long long)a * K0 + (unsigned long long)b * K1 +
                           (unsigned long long)c * K2 + K4;
    z = (z ^ (z >> 30)) * K1;
    z = (z ^ (z >> 27)) * K2;
    return z ^ (z >> 31);
}

__device__ inline long long octave(unsigned long long seed, int o, long long x, long long y, int lg)
{
    const long long s = 1ll << lg;
    const long long X = x >> lg, Y = y >> lg, fx = x & (s - 1), fy = y & (s - 1);
    const long long a = (long long)(mix(seed, 100 + o, X, Y) & 255), b = (long long)(mix(seed, 100 + o, X + 1, Y) & 255);
    const long long c = (long long)(mix(seed, 100 + o, X, Y + 1) & 255), d = (long long)(mix(seed, 100 + o, X + 1, Y + 1) & 255);
    const long long v = a * (s - fx) * (s - fy) + b * fx * (s - fy) + c * (s - fx) * fy + d * fx * fy;
    return v >> (2 * lg);
}

__device__ inline int left_value(unsigned long long seed, long long x, long long y)
{
    x += 4096; y += 4096;
    const long long v = 2 * octave(seed, 0, x, y, 4) + 2 * octave(seed, 1, x, y, 3) + 2 * octave(seed, 2, x, y, 2) +
                        octave(seed, 3, x, y, 1) + octave(seed, 4, x, y, 0);
    return (int)(v >> 3);
}

__global__ void k_synth_params(SynthParams* P, unsigned long long seed0, int first, int n, int W, int H, int D)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    SynthParams p;
    const unsigned long long seed = seed0 + (unsigned long long)(first + f);
    p.seed = seed;
    p.lo = 2; p.hi = max(3, D - 3);
    p.nrect = 3 + (int)(mix(seed, 1, 0, 0) % 4ull);
    for (int k = 0; k < p.nrect; ++k) {
        int h[5];
        for (int t = 0; t < 5; ++t) h[t] = (int)(mix(seed, 2, k, t) % (1ull << 20));
        const int rw = max(8, W / 8 + h[0] % max(1, W / 4));
        const int rh = max(8, H / 8 + h[1] % max(1, H / 4));
        const int x0 = h[2] % max(1, W - rw), y0 = h[3] % max(1, H - rh);
        p.rect[k][0] = x0; p.rect[k][1] = y0; p.rect[k][2] = x0 + rw; p.rect[k][3] = y0 + rh;
        p.rect[k][4] = p.lo + h[4] % (p.hi - p.lo + 1);
    }
    p.base = p.lo + (int)(mix(seed, 3, 0, 0) % (unsigned long long)max(1, (p.hi - p.lo) / 2));
    p.gx = (int)(mix(seed, 3, 1, 0) % 17ull);
    p.gy = (int)(mix(seed, 3, 2, 0) % 33ull);
    p.nout = (W <= 16 || H <= 16) ? 0 : 20;
    for (int k = 0; k < p.nout; ++k) {
        int h[3];
        for (int t = 0; t < 3; ++t) h[t] = (int)(mix(seed, 5, k, t) % (1ull << 20));
        p.outl[k][0] = 4 + h[0] % (W - 12); p.outl[k][1] = 4 + h[1] % (H - 12);
        p.outl[k][2] = p.lo + h[2] % (p.hi - p.lo + 1);
    }
    p.ps = max(4, min(64, min(W / 4, H / 4)));
    p.fx = (int)(mix(seed, 6, 0, 0) % (unsigned long long)max(1, W - p.ps));
    p.fy = (int)(mix(seed, 6, 1, 0) % (unsigned long long)max(1, H - p.ps));
    p.sx = (int)(mix(seed, 7, 0, 0) % (unsigned long long)max(1, W - p.ps));
    p.sy = (int)(mix(seed, 7, 1, 0) % (unsigned long long)max(1, H - p.ps));
    P[f] = p;
}

__global__ __launch_bounds__(256) void k_synth(const SynthParams* P, Plane8W L, Plane8W R, int W, int H)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, f = blockIdx.z;
    const SynthParams& p = P[f];
    const unsigned long long seed = p.seed;
    int lv, rv;
    const bool in_stripe = x >= p.sx && x < p.sx + p.ps && y >= p.sy && y < p.sy + p.ps;
    const bool in_flat = x >= p.fx && x < p.fx + p.ps && y >= p.fy && y < p.fy + p.ps;
    if (in_stripe) {
        lv = rv = ((x >> 2) & 1) ? 200 : 60;
    } else if (in_flat) {
        lv = rv = 100;
    } else {
        lv = left_value(seed, x, y);
        int dalt = -1;
        for (int k = p.nout - 1; k >= 0; --k)
            if (x >= p.outl[k][0] && x < p.outl[k][0] + 5 && y >= p.outl[k][1] && y < p.outl[k][1] + 5) { dalt = p.outl[k][2]; break; }
        if (dalt >= 0) {
            rv = left_value(seed, x + dalt, y);
        } else {
            int d = p.base + (int)(((long long)p.gx * x + (long long)p.gy * y) >> 10);
            d = min(max(d, p.lo), p.hi);
            for (int k = 0; k < p.nrect; ++k)
                if (x >= p.rect[k][0] && x < p.rect[k][2] && y >= p.rect[k][1] && y < p.rect[k][3]) d = p.rect[k][4];
            rv = left_value(seed, x + d, y);
            rv += (int)(mix(seed, 4, x, y) % 5ull) - 2;
            rv = min(max(rv, 0), 255);
        }
    }
    L.base[(size_t)f * L.frame + (size_t)y * L.pitch + x] = (uint8_t)lv;
    R.base[(size_t)f * R.frame + (size_t)y * R.pitch + x] = (uint8_t)rv;
}

size_t synth_scratch_bytes(int n) { return sizeof(SynthParams) * (size_t)n; }

void launch_synth(uint64_t seed, int first_frame, int n, int W, int H, int D, Plane8W L, Plane8W R,
                  void* param_scratch, hipStream_t stream)
{
    SynthParams* P = (SynthParams*)param_scratch;
    hipLaunchKernelGGL(k_synth_params, dim3((n + 63) / 64), dim3(64), 0, stream, P, (unsigned long long)seed, first_frame, n, W, H, D);
    hipLaunchKernelGGL(k_synth, dim3((W + 255) / 256, H, n), dim3(256), 0, stream, P, L, R, W, H);
}

}  // namespace rtdm
