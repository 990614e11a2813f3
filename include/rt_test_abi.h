/* rt_test_abi.h -- test-only entry points (round 5: out of the product library).
 *
 * The product, ray_tracer_2_amd/librt2_mi355x.so, exports include/rt_abi.h and nothing else.  The entry points below
 * exist only in ray_tracer_2_amd/librt2_mi355x_test.so: the SAME sources and compile flags plus -DRT_TEST_ENTRIES=1
 * (ray_tracer_2_amd/build.py: build_test_library), which also exports every symbol of rt_abi.h, so that a test can
 * drive a handle and these entry points through one library.  tests/test_abi.py checks both export lists.
 */
#ifndef RT_TEST_ABI_H
#define RT_TEST_ABI_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test-only: the device's evaluation of the kernels' arithmetic building blocks, element-wise over
 * host arrays of n floats (bit patterns for integer inputs/outputs).  fn: 0 log, 1 cos, 2 sin, 3 exp,
 * 4 exp2, 5 log2, 6 pow(x,y), 7 acos, 8 atan2(x,y), 9 sqrt, 10 x/y, 11 rand() from RNG state bits x,
 * 12 next_random_number of state x (bits), 13 trig_signbits(x) (bits), 14 rand_normal_dist() from
 * state x, 15 f32(u32 bits x) * 2^-32, 16 normalize(x, y, x*y).x, 17 the kernels' reciprocal rcp_(x), 18 their
 * sqrt_dev(x).  The second call filters one RGBA8 sRGB texture at n (u, v) pairs (wgsl:455 as csrc/rt_texture.h defines
 * it).  The third compares the kernels' short reciprocal (which = 0) / square root (which = 1) with the compiler's IEEE
 * 1.0f / x / sqrt on the device for EVERY float in the range the short form serves, and the sky's three shortcuts
 * (which = 2, 3, 4: wgsl:215-218 with the branches of smoothstep / pow taken apart) with their literal forms for every
 * float in [-1.5, 1.5] / [0, 1.5]: out3 = {floats checked, mismatches, a mismatching bit pattern}. */
int rt_test_device_units(rt_handle* h, int fn, const float* x, const float* y, float* out, uint64_t n);
int rt_test_sweep(rt_handle* h, int which, uint64_t* out3);
int rt_test_device_sample_texture(rt_handle* h, const rt_texture_desc* tex, const float* uv, float* rgba_out,
                                  uint64_t n);

/* Test-only: raw copy of a buffer of the last wavefront sequence (which: 0 path state, 1 hit records, 2 the two slot
 * lists, 3 the per-round list counts; layouts in csrc/rt_device.h), or (which = 4) the pixels parked in front of each round
 * of the last deferred-walk sequence (72 u32), or (which = 5, 6) the park records of its even / odd rounds. */
int rt_test_read_wavefront(rt_handle* h, int which, void* out, uint64_t bytes);
/* The grouped ncclSend / ncclRecv gather of rt_render_multi against the RCCL-shaped library at `lib_path`, on fake
 * buffers and without any HIP call (runs without a GPU): checks that a failure inside the group still closes the group
 * and aborts the communicators.  Error text: rt_last_error(NULL). */
int rt_test_rccl_gather(const char* lib_path, int n_ranks);
/* The automatic depth of option "frame_ahead" for a one-frame call that continues an accumulation (no GPU needed): the
 * scene staged in LDS (1) or read from global memory (0), the texels of the call's share, samples per pixel, bounces,
 * and whether the host counts as one that waits for every frame.  0 = the call renders its own frame only. */
int rt_test_frame_ahead_depth(int lds_scene, uint64_t texels, int rays_per_pixel, int number_of_bounces, int host_waits);

#ifdef __cplusplus
}
#endif

#endif
