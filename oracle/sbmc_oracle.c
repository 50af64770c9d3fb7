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

/* the operators (sbmc_oracle_ops.inc), once in the reference's float32 ... */
#define REAL float
#define FMA fmaf
#define NAME(op) sbmc_oracle_##op
#include "sbmc_oracle_ops.inc"
#undef REAL
#undef FMA
#undef NAME
/* ... and once in float64 (sbmc_oracle_<op>_f64): not the reference's arithmetic, the tests' yardstick */
#define REAL double
#define FMA fma
#define NAME(op) sbmc_oracle_##op##_f64
#include "sbmc_oracle_ops.inc"
#undef REAL
#undef FMA
#undef NAME


int sbmc_oracle_abi_version(void) { return 2; }
