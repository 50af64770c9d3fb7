/*
 * sbmc_oracle.c -- CPU restatement of the three SBMC native operators.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle (and the timed
 * "port" CPU baseline of bench.py); nothing under sbmc_amd/ may import, link
 * or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * What it restates (citations relative to /root/reference):
 *   - kernel_weighting       src/kernel_weighting.cpp:28-64
 *   - kernel_weighting_grad  src/kernel_weighting.cpp:68-124
 *   - scatter2gather         src/scatter2gather.cpp:29-52
 * The reference implements these as Halide *generators* (Halide v8.0.0,
 * dockerfiles/cuda-sbmc.dockerfile:93).  Halide is a third-party dependency
 * that is absent from this image, so the reference's own CPU binaries cannot
 * be produced (unbuildable: needs Halide.h / libHalide / GenGen.cpp); the
 * algorithm text however is entirely in the two generator files, and this file
 * follows it expression by expression.
 *
 * Pinning: the restatement is checked against every analytic known-answer
 * test the reference holds for these ops (tests/test_functions.py:43-70,
 * 72-103, 105-144, 164-185, 187-208 and tests/test_modules.py:63-99,102-140),
 * re-stated in tests/test_oracle_kats.py, and -- in the authoring container --
 * by running the reference's *own unmodified test files* on top of it
 * (oracle/pin_against_reference.py).
 *
 * Layout (torch order == reversed Halide order, sbmc/functions.py:46,79):
 *   data     [bs, c,  h, w]      Halide (x, y, ci, n)
 *   weights  [bs, kh, kw, h, w]  Halide (x, y, dx, dy, n)
 *   sum_w    [bs, h, w]          Halide (x, y, n)
 *
 * Accumulation order: Halide places the reduction domain innermost for an
 * update definition, r.x (kernel column) fastest, then r.y; every output
 * element starts from 0.0f (summed(...) = 0.0f; summed(...) += w * h).  The
 * loops below keep exactly that per-element order (ry outer, rx inner) while
 * iterating x innermost so gcc can vectorise across outputs, which is what the
 * reference CPU schedule does too (vectorize(x, 8), kernel_weighting.cpp:165-187).
 * Multiply-add is a fused fmaf: Halide's LLVM backend allows FP contraction on
 * FMA-capable hosts.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, c, y, x, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (y)) * (W) + (x))

/* kernel_weighting.cpp:28-64
 *   summed(x,y,c,n) = 0; summed += w(x,y,r.x,r.y,n) * homogeneous(x+r.x-(kw-1)/2, y+r.y-(kh-1)/2, c, n)
 *   homogeneous = c < channels ? data_zero_padded : 1.0f          (:49)
 *   output = summed[c < channels]; sum_w = summed[c == channels]   (:56-57)
 * sum_w is therefore NOT masked at the image boundary. */
int sbmc_oracle_kernel_weighting(const float *data, const float *weights,
                                 float *output, float *sum_w,
                                 int bs, int c, int h, int w, int kh, int kw)
{
    if (bs < 0 || c < 0 || h < 0 || w < 0 || kh <= 0 || kw <= 0) return 1;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const long rows = (long)bs * h;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(c + 1) * (w > 0 ? w : 1));
#pragma omp for schedule(dynamic, 8)
        for (long row = 0; row < rows; ++row) {
            const int n = (int)(row / h), y = (int)(row % h);
            memset(acc, 0, sizeof(float) * (size_t)(c + 1) * w);
            for (int ry = 0; ry < kh; ++ry) {
                const int ys = y + ry - ph;
                const int yin = (ys >= 0 && ys < h);
                for (int rx = 0; rx < kw; ++rx) {
                    const float *wrow = weights + IDX4(n, ry * kw + rx, y, 0, kh * kw, h, w);
                    const int xo = rx - pw;
                    /* x range whose source column x+xo is inside the image */
                    int x0 = xo < 0 ? -xo : 0;
                    int x1 = xo > 0 ? w - xo : w;
                    if (x1 < x0) x1 = x0;
                    if (yin) {
                        for (int ch = 0; ch < c; ++ch) {
                            const float *drow = data + IDX4(n, ch, ys, 0, c, h, w);
                            float *a = acc + (size_t)ch * w;
                            /* outside [x0,x1): data is 0 -> fmaf(w, 0, a) == a */
                            for (int x = x0; x < x1; ++x)
                                a[x] = fmaf(wrow[x], drow[x + xo], a[x]);
                        }
                    }
                    float *aw = acc + (size_t)c * w;
                    for (int x = 0; x < w; ++x)
                        aw[x] = fmaf(wrow[x], 1.0f, aw[x]);
                }
            }
            for (int ch = 0; ch < c; ++ch)
                memcpy(output + IDX4(n, ch, y, 0, c, h, w), acc + (size_t)ch * w, sizeof(float) * w);
            memcpy(sum_w + ((size_t)n * h + y) * w, acc + (size_t)c * w, sizeof(float) * w);
        }
        free(acc);
    }
    return 0;
}

/* kernel_weighting.cpp:68-124
 *   d_data(x,y,c,n)  = sum_r  Wz(x+r.x-pw, y+r.y-ph, kw-1-r.x, kh-1-r.y, n) * dOz(x+r.x-pw, y+r.y-ph, c, n)   (:93-105)
 *   d_weights(x,y,dx,dy,n) = d_sum_w(x,y,n) + sum_ch Dz(x+dx-pw, y+dy-ph, ch, n) * dO(x,y,ch,n)               (:111-117)
 * The sum_w input is unused by the reference algorithm (kept for signature parity). */
int sbmc_oracle_kernel_weighting_grad(const float *data, const float *weights,
                                      const float *sum_w, const float *d_output,
                                      const float *d_sum_w, float *d_data,
                                      float *d_weights,
                                      int bs, int c, int h, int w, int kh, int kw)
{
    (void)sum_w;
    if (bs < 0 || c < 0 || h < 0 || w < 0 || kh <= 0 || kw <= 0) return 1;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;

    /* d_data */
    const long drows = (long)bs * c * h;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(w > 0 ? w : 1));
#pragma omp for schedule(dynamic, 8)
        for (long row = 0; row < drows; ++row) {
            const int y = (int)(row % h);
            const int ch = (int)((row / h) % c);
            const int n = (int)(row / ((long)h * c));
            memset(acc, 0, sizeof(float) * w);
            for (int ry = 0; ry < kh; ++ry) {
                const int ys = y + ry - ph;
                if (ys < 0 || ys >= h) continue; /* Wz and dOz both 0 */
                for (int rx = 0; rx < kw; ++rx) {
                    const int xo = rx - pw;
                    int x0 = xo < 0 ? -xo : 0;
                    int x1 = xo > 0 ? w - xo : w;
                    const float *wrow = weights + IDX4(n, (kh - 1 - ry) * kw + (kw - 1 - rx), ys, 0, kh * kw, h, w);
                    const float *grow = d_output + IDX4(n, ch, ys, 0, c, h, w);
                    for (int x = x0; x < x1; ++x)
                        acc[x] = fmaf(wrow[x + xo], grow[x + xo], acc[x]);
                }
            }
            memcpy(d_data + IDX4(n, ch, y, 0, c, h, w), acc, sizeof(float) * w);
        }
        free(acc);
    }

    /* d_weights */
    const long wrows = (long)bs * kh * kw * h;
#pragma omp parallel for schedule(dynamic, 8)
    for (long row = 0; row < wrows; ++row) {
        const int y = (int)(row % h);
        const int tap = (int)((row / h) % ((long)kh * kw));
        const int n = (int)(row / ((long)h * kh * kw));
        const int dy = tap / kw, dx = tap % kw;
        const int ys = y + dy - ph, xo = dx - pw;
        float *out = d_weights + IDX4(n, tap, y, 0, kh * kw, h, w);
        const float *dsw = d_sum_w + ((size_t)n * h + y) * w;
        for (int x = 0; x < w; ++x) out[x] = dsw[x];
        if (ys < 0 || ys >= h) continue;
        int x0 = xo < 0 ? -xo : 0;
        int x1 = xo > 0 ? w - xo : w;
        for (int ch = 0; ch < c; ++ch) {
            const float *drow = data + IDX4(n, ch, ys, 0, c, h, w);
            const float *grow = d_output + IDX4(n, ch, y, 0, c, h, w);
            for (int x = x0; x < x1; ++x)
                out[x] = fmaf(drow[x + xo], grow[x], out[x]);
        }
    }
    return 0;
}

/* scatter2gather.cpp:29-52
 *   output(x,y,dx,dy,n) = Wz(x+dx-pw, y+dy-ph, kw-1-dx, kh-1-dy, n) */
int sbmc_oracle_scatter2gather(const float *weights, float *output,
                               int bs, int h, int w, int kh, int kw)
{
    if (bs < 0 || h < 0 || w < 0 || kh <= 0 || kw <= 0) return 1;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const long rows = (long)bs * kh * kw * h;
#pragma omp parallel for schedule(dynamic, 16)
    for (long row = 0; row < rows; ++row) {
        const int y = (int)(row % h);
        const int tap = (int)((row / h) % ((long)kh * kw));
        const int n = (int)(row / ((long)h * kh * kw));
        const int dy = tap / kw, dx = tap % kw;
        const int ys = y + dy - ph, xo = dx - pw;
        float *out = output + IDX4(n, tap, y, 0, kh * kw, h, w);
        if (ys < 0 || ys >= h) {
            memset(out, 0, sizeof(float) * w);
            continue;
        }
        const float *src = weights + IDX4(n, (kh - 1 - dy) * kw + (kw - 1 - dx), ys, 0, kh * kw, h, w);
        for (int x = 0; x < w; ++x) {
            const int xs = x + xo;
            out[x] = (xs >= 0 && xs < w) ? src[xs] : 0.0f;
        }
    }
    return 0;
}

int sbmc_oracle_abi_version(void) { return 1; }
