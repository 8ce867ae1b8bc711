// k_misc.hip -- K9: pointwise helpers (accumulate, fill, channels-last <-> planar).
#include "fr3d_internal.h"

namespace fr3d {

__global__ void __launch_bounds__(256) k_axpy(float *__restrict__ y, const float *__restrict__ x, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = y[t] + x[t];
}

// read-only stream for fr3d_stream_probe: every thread sums one element of each of eight streams (the
// SOR sweep is read-heavy, 95 of its 115 real bytes per update are reads); one store per wave
__global__ void __launch_bounds__(256) k_read8(const float *__restrict__ x, long long n, float *__restrict__ sink)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float a = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; q++) a += x[(size_t)q * n + t];
    if (a == 123456.789f) sink[t & 63] = a;  // never true for the zero-filled probe buffer; keeps the loads alive
}
void launch_read8(hipStream_t st, const float *x, long long n, float *sink)
{
    if (n > 0) hipLaunchKernelGGL(k_read8, dim3(cdiv(n, 256)), dim3(256), 0, st, x, n, sink);
    FR3D_LAUNCH_CHECK();
}

__global__ void __launch_bounds__(256) k_fill(float *__restrict__ y, float v, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = v;
}

__global__ void __launch_bounds__(256)
k_pack(const float *__restrict__ planar, int C, long long n, float *__restrict__ inter)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    long long v = t / C;
    int c = (int)(t - v * C);
    inter[t] = planar[(size_t)c * n + v];
}

__global__ void __launch_bounds__(256)
k_unpack(const float *__restrict__ inter, int C, long long n, float *__restrict__ planar)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    int c = (int)(t / n);
    long long v = t - (long long)c * n;
    planar[t] = inter[(size_t)v * C + c];
}

// f-4 update_reference (compensate_recording_3D.py:395-429): the fp64 stack mean, volume by volume in
// stack order like np.mean(axis=0): acc = x0, acc += x1, ... ; out = acc / n
__global__ void __launch_bounds__(256)
k_accum_f64(double *__restrict__ acc, const float *__restrict__ x, long long n, int first)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) acc[t] = first ? (double)x[t] : acc[t] + (double)x[t];
}
__global__ void __launch_bounds__(256)
k_mean_store(const double *__restrict__ acc, long long n, int C, int c, double count, double *__restrict__ out)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[(size_t)t * C + c] = acc[t] / count;
}
// np.mean(stack, axis=0) of a float32 stack (count, n) as NumPy evaluates it: the volumes are added in
// stack order in float32, the sum is divided by float32(count) (the rolling w_init of the batch driver,
// compensate_recording_3D.py:481-485 and :342-393)
__global__ void __launch_bounds__(256)
k_mean_stack_f32(const float *__restrict__ stack, int count, long long n, float *__restrict__ out)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float acc = stack[t];
    for (int q = 1; q < count; q++) acc = acc + stack[(size_t)q * n + t];
    out[t] = acc / (float)count;
}
void launch_mean_stack_f32(hipStream_t st, const float *stack, int count, long long n, float *out)
{
    if (n > 0 && count > 0) hipLaunchKernelGGL(k_mean_stack_f32, dim3(cdiv(n, 256)), dim3(256), 0, st, stack, count, n, out);
    FR3D_LAUNCH_CHECK();
}

void launch_accum_f64(hipStream_t st, double *acc, const float *x, long long n, bool first)
{
    if (n > 0) hipLaunchKernelGGL(k_accum_f64, dim3(cdiv(n, 256)), dim3(256), 0, st, acc, x, n, first ? 1 : 0);
    FR3D_LAUNCH_CHECK();
}
void launch_mean_store(hipStream_t st, const double *acc, long long n, int C, int c, double count, double *out)
{
    if (n > 0) hipLaunchKernelGGL(k_mean_store, dim3(cdiv(n, 256)), dim3(256), 0, st, acc, n, C, c, count, out);
    FR3D_LAUNCH_CHECK();
}

void launch_axpy(hipStream_t st, float *y, const float *x, long long n)
{
    if (n > 0) hipLaunchKernelGGL(k_axpy, dim3(cdiv(n, 256)), dim3(256), 0, st, y, x, n);
    FR3D_LAUNCH_CHECK();
}
void launch_fill(hipStream_t st, float *y, float v, long long n)
{
    if (n > 0) hipLaunchKernelGGL(k_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, y, v, n);
    FR3D_LAUNCH_CHECK();
}
// three separate component arrays -> (n, 3) interleaved; one voxel per thread (a wave reads three 256-B runs and
// writes one 768-B run)
__global__ void __launch_bounds__(256)
k_pack3(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c, long long n,
        float *__restrict__ inter)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float x = a[t], y = b[t], z = c[t];
    float *o = inter + t * 3;
    o[0] = x;
    o[1] = y;
    o[2] = z;
}
void launch_pack3(hipStream_t st, const float *a, const float *b, const float *c, long long n, float *inter)
{
    if (n > 0) hipLaunchKernelGGL(k_pack3, dim3(cdiv(n, 256)), dim3(256), 0, st, a, b, c, n, inter);
    FR3D_LAUNCH_CHECK();
}
void launch_pack(hipStream_t st, const float *planar, int C, long long n, float *inter)
{
    if (n > 0) hipLaunchKernelGGL(k_pack, dim3(cdiv(n * C, 256)), dim3(256), 0, st, planar, C, n, inter);
    FR3D_LAUNCH_CHECK();
}
void launch_unpack(hipStream_t st, const float *inter, int C, long long n, float *planar)
{
    if (n > 0) hipLaunchKernelGGL(k_unpack, dim3(cdiv(n * C, 256)), dim3(256), 0, st, inter, C, n, planar);
    FR3D_LAUNCH_CHECK();
}

}  // namespace fr3d

// ---- f-2: per-volume flow statistics (motion_correction/compensate_recording_3D.py:488-508) ----
// |w| mean and max, mean divergence (np.gradient: central differences, one-sided at the ends,
// unit spacing) and mean u,v,w of a (Z,Y,X,3) fp32 flow.  Per-voxel values are formed in fp32
// exactly like NumPy does on the float32 flow; sums are accumulated in fp64 per block and finished
// on the host (deterministic, no atomics).
namespace fr3d {

__global__ void __launch_bounds__(256)
k_flow_stats(const float *__restrict__ flow, int Z, int Y, int X, double *__restrict__ partial)
{
    const long long n = (long long)Z * Y * X;
    double s_mag = 0, s_div = 0, s_u = 0, s_v = 0, s_w = 0;
    float m_mag = 0.0f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % X);
        const long long r = e / X;
        const int y = (int)(r % Y), z = (int)(r / Y);
        const float u = flow[e * 3 + 0], v = flow[e * 3 + 1], w = flow[e * 3 + 2];
        const float mag = sqrtf(u * u + v * v + w * w);
        s_mag += (double)mag;
        m_mag = fmaxf(m_mag, mag);
        s_u += (double)u; s_v += (double)v; s_w += (double)w;
        auto grad = [&](int pos, int len, long long stride, int comp) -> float {
            if (len == 1) return 0.0f;  // np.gradient needs >= 2 samples; the pipeline never has 1
            const float *p = flow + e * 3 + comp;
            if (pos == 0) return (p[stride * 3] - p[0]) / 1.0f;
            if (pos == len - 1) return (p[0] - p[-stride * 3]) / 1.0f;
            return (p[stride * 3] - p[-stride * 3]) / 2.0f;
        };
        const float dux = grad(x, X, 1, 0), dvy = grad(y, Y, X, 1), dwz = grad(z, Z, (long long)Y * X, 2);
        s_div += (double)((dux + dvy) + dwz);
    }
    __shared__ double sh[6][256];
    sh[0][threadIdx.x] = s_mag; sh[1][threadIdx.x] = (double)m_mag; sh[2][threadIdx.x] = s_div;
    sh[3][threadIdx.x] = s_u; sh[4][threadIdx.x] = s_v; sh[5][threadIdx.x] = s_w;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                if (q == 1) sh[q][threadIdx.x] = fmax(sh[q][threadIdx.x], sh[q][threadIdx.x + off]);
                else sh[q][threadIdx.x] += sh[q][threadIdx.x + off];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[(size_t)blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x][0];
}

void launch_flow_stats(hipStream_t st, const float *flow, int Z, int Y, int X, int nblocks, double *partial)
{
    hipLaunchKernelGGL(k_flow_stats, dim3(nblocks), dim3(256), 0, st, flow, Z, Y, X, partial);
    FR3D_LAUNCH_CHECK();
}

}  // namespace fr3d

// Which XCD each workgroup of a grid_x x grid_y launch runs on (HW_REG_XCC_ID, bits 3:0): a diagnostic for the
// placement assumption behind the sweep's XCD-aware tile order (ids with equal blockIdx.x % 8 share an XCD).
namespace fr3d {
__global__ void k_xcd_probe(int *__restrict__ out)
{
    if (threadIdx.x == 0) {
        const unsigned v = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // hwreg(HW_REG_XCC_ID), all 32 bits
        out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (int)(v & 0xf);
    }
}
void launch_xcd_probe(hipStream_t st, int gx, int gy, int *out)
{
    hipLaunchKernelGGL(k_xcd_probe, dim3(gx, gy), dim3(128), 0, st, out);
    FR3D_LAUNCH_CHECK();
}
}  // namespace fr3d
