/* include/rtk.h — C ABI of librtk.so, the MI355X (gfx950) replacement for the reference's
 * "Image Generation" scope.
 *
 * The reference (cugone/RayTracingInOneWeekend, branch master) has NO plugin / operator / FFI
 * interface.  The seam this library fills is the block guarded by
 * PROFILE_LOG_SCOPE("Image Generation"), RayTracingInOneWeekend/main.cpp:62-78: its inputs are the
 * already-built HittableList (main.cpp:45), the Camera (main.cpp:54) and the four CLI integers
 * (main.cpp:24-41); its output is the per-pixel colour sum handed to write_color_binary
 * (main.cpp:74, Color.hpp:28-48).  Each entry point below cites the reference code it replaces.
 * File:line citations are relative to /root/reference/RayTracingInOneWeekend/.
 *
 * Conventions: plain C, POD structs, caller owns every host pointer for the duration of a call,
 * the library owns all device memory of a context.  Every function returns RTK_OK (0) or a negative
 * rtk_status; nothing throws.  rtk_last_error() returns a thread-local message for the last failure.
 * A context is used by one host thread at a time.  There is no CPU fallback: without a usable HIP
 * device rtk_context_create fails with RTK_ERR_NO_DEVICE.
 */
#ifndef RTK_H
#define RTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtk_status {
    RTK_OK = 0,
    RTK_ERR_BAD_ARG = -1,    /* null pointer, non-positive size, tile not 64 pixels, ... */
    RTK_ERR_NO_DEVICE = -2,  /* no HIP device / device index out of range */
    RTK_ERR_HIP = -3,        /* a HIP runtime call failed; see rtk_last_error() */
    RTK_ERR_NO_SCENE = -4,   /* render/intersect before rtk_set_scene */
    RTK_ERR_TOO_LARGE = -5   /* size beyond a documented limit */
} rtk_status;

/* Material::Type, Material.hpp:10-15 (enum order). */
enum { RTK_MAT_NONE = 0, RTK_MAT_LAMBERTIAN = 1, RTK_MAT_METAL = 2, RTK_MAT_GLASS = 3 };

/* Flattened HittableList of Sphere3 (HittableList.hpp:27, Sphere3.hpp:22-24, Material.hpp:19-24),
 * structure-of-arrays, LIST ORDER PRESERVED (the closest-hit tie rule depends on it,
 * HittableList.hpp:34-38 with Sphere3.hpp:43).  `color` is Material::color, the multiplier
 * ray_color uses (main.cpp:93); Material::attenuation is never read on the path and is not carried. */
typedef struct rtk_scene {
    int32_t n;
    const float* cx;
    const float* cy;
    const float* cz;
    const float* radius;
    const float* mat_r;
    const float* mat_g;
    const float* mat_b;
    const float* roughness;
    const float* metallic;
    const float* ior;          /* Material::refractionIndex */
    const int32_t* mat_type;   /* RTK_MAT_* */
} rtk_scene;

/* The eight private fields of Camera (Camera.hpp:32-39) as the constructor leaves them
 * (Camera.hpp:8-23). */
typedef struct rtk_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float w[3];
    float lens_radius;
} rtk_camera;

/* Inputs of the pixel/sample loops, main.cpp:24-41 and 64-76. */
typedef struct rtk_params {
    int32_t width;        /* image_width  (2 .. 65535) */
    int32_t height;       /* image_height (2 .. 65535) */
    int32_t spp;          /* samples_per_pixel */
    int32_t max_depth;    /* max_depth */
    uint64_t seed;        /* key of the counter-based RNG that replaces MathUtils.hpp:15-25 */
    float t_min;          /* 0.001f at main.cpp:90 */
    int32_t rank;         /* this process renders tiles i with i % world_size == rank ... */
    int32_t world_size;   /* ... of the row-major grid of tile_w x tile_h pixel tiles */
    int32_t tile_w;       /* tile_w * tile_h must be 64 (one wavefront of pixels); 0,0 -> 8 x 8 */
    int32_t tile_h;
    int32_t run_len;      /* how a pixel's samples are cut into units of work; the IMAGE does not depend on it (every sample
                           * becomes 32.32 fixed point before it is summed).  0: the default decreasing partition, see
                           * rtk_run_schedule; > 0: runs of this many samples; < 0: decreasing with k = -run_len */
    int32_t sample_offset; /* index of the first sample (default 0): a frame can be rendered progressively, in batches
                           * [offset, offset + spp) whose sums the caller adds; batches never share RNG counters */
} rtk_params;

/* What one render did.  `rays` counts ray SEGMENTS = calls of HittableList::hit (main.cpp:90),
 * primary rays included, counted by the kernel itself. */
typedef struct rtk_stats {
    uint64_t rays;
    uint64_t samples;
    uint64_t sphere_tests;     /* list walk: rays x scene size (the filter tests every sphere for every segment);
                                * uniform grid: the exact tests of the cells the rays crossed */
    uint64_t exact_tests;      /* candidates re-tested with the reference's exact arithmetic */
    uint64_t lane_slots;       /* 64 x wavefront passes through the intersection loop */
    float kernel_ms;           /* HIP-event time of the render kernel alone, summed over `frames_timed` frames */
    float total_ms;            /* clear + render + finalise of the last frame (only from a call given `stats`) */
    uint32_t frames;           /* frames rendered since the last rtk_reset_stats */
    uint32_t frames_timed;     /* frames whose kernel time is in kernel_ms (at most 256 are kept) */
    uint64_t max_wave_passes;  /* most passes any one wavefront made in one frame: the kernel's critical path
                                * (lane_slots / 64 / waves is the mean; their ratio is the load balance across waves) */
    uint64_t waves;            /* wavefronts of the last render kernel launch */
    float drain_frac;          /* last frame: share of the kernel's duration that passed after the FIRST wavefront ran out of work
                                * (the tail during which the GPU empties; fixed 100 MHz device clock) */
    float reserved_;
} rtk_stats;

typedef struct rtk_context rtk_context;

/* Number of visible HIP devices (0 when there is none). */
int rtk_device_count(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* rtk_last_error(void);

/* Persistent state on one device: stream, scene buffers, accumulators.  Replaces nothing in the
 * reference (which has no device); it exists so that frames after the first pay no allocation. */
int rtk_context_create(int device, rtk_context** out);
void rtk_context_destroy(rtk_context* ctx);

/* Use an existing HIP stream (a hipStream_t passed as void*) for all work of the context;
 * NULL selects the context's own stream.  For callers that own device memory, e.g. torch.
 * A switch is ordered: work submitted to the new stream waits for what the context queued on the old one (its
 * accumulators, packet queue and run table are shared by consecutive frames).  Lifetime: the call does not touch the old
 * stream (the event it waits on was recorded when the work was queued), so a caller may destroy its stream once the work it
 * queued there is complete or synchronised, and switch afterwards. */
int rtk_context_set_stream(rtk_context* ctx, void* hip_stream);

/* How the closest hit is found.  No counterpart in the reference, whose HittableList::hit (HittableList.hpp:30-42)
 * walks the whole list for every ray; SURVEY.md §8f-3.
 *   RTK_ACCEL_LIST (default): the list walk — every ray meets every sphere; the path the roofline is quoted on.
 *   RTK_ACCEL_GRID: a uniform grid over the spheres, built by the library at rtk_set_scene / at this call; a ray
 *     tests only the spheres registered in the cells it crosses.  Results (hit, index, t, images, ray counts) are
 *     bit-identical to the list walk; rays the grid's margins were not sized for (non-finite, origins farther than
 *     3/4 of the scene's diagonal from its centre) take the list walk inside the same kernel.  Meant for lists far beyond
 *     the 1024 spheres that fit in LDS (BASELINE.json config 4).
 * A scene that cannot be put in a grid (a non-finite sphere: the reference's result then depends on the LIST order)
 * is rendered with the list walk; rtk_accel_in_use tells which one the next call will take and rtk_last_error()
 * holds the reason after rtk_context_set_accel / rtk_set_scene. */
enum { RTK_ACCEL_LIST = 0, RTK_ACCEL_GRID = 1 };
int rtk_context_set_accel(rtk_context* ctx, int32_t mode);
int rtk_accel_in_use(rtk_context* ctx);

/* How the resident scene is laid out for the kernels (diagnostics for bench.py and the profile summaries; no counterpart
 * in the reference): out4 = { spheres, CLASS words (32 consecutive list entries whose filterable spheres share cy and
 * radius — every small sphere of random_scene(), main.cpp:124, qualifies — so that their filter test is 15 terms of one
 * dot product: two v_mfma_f32_32x32x16_f16 per word of 32 spheres x 64 rays instead of four for a general word),
 * 1 if the scene is LDS-resident (at most 1024 spheres), LDS bytes per workgroup of the render kernel }.  For a larger scene
 * the second number counts the words that have a tile in their own frame (0: the scene streams through the vector-ALU
 * filter; otherwise the matrix-pipe filter runs per word in word-local coordinates). */
int rtk_scene_layout(rtk_context* ctx, int32_t* out4);

/* Upload the flattened world.  Replaces the traversal state HittableList::hit walks
 * (HittableList.hpp:27,34): precomputes, bit-identically to the reference's own fp32 operations,
 * radius*radius (Sphere3.hpp:35) and 1/radius (Vector3.cpp:35-37 via Sphere3.hpp:54). */
int rtk_set_scene(rtk_context* ctx, const rtk_scene* scene);

/* Number of tiles rank `rank` of `world_size` owns for a width x height image. */
int64_t rtk_local_tile_count(int32_t width, int32_t height, int32_t tile_w, int32_t tile_h, int32_t rank,
                             int32_t world_size);

/* The partition of the spp samples of a pixel into runs that a rank owning width x height pixels uses for
 * (spp, run_len): n pairs (first sample, length) into out_pairs (2 * cap uint32), returns n (out_pairs may be NULL to
 * query n).  Pure host function (no device needed).  A rank of a sharded frame sizes the partition for ITS pixel count
 * (finer when it owns fewer pixels, for load balance); images stay bit-identical across GPU counts because samples are
 * summed as integers, whatever the partition.  No counterpart in the reference (its sample loop is serial, main.cpp:68). */
int rtk_run_schedule(int32_t spp, int32_t width, int32_t height, int32_t run_len, uint32_t* out_pairs, int32_t cap);

/* main.cpp:64-76 for the tiles of p->rank: per-pixel fp32 colour sums over spp samples, written to
 * DEVICE memory `d_tile_sums` as [local tile][tile_h][tile_w][3] floats (local tile k is global tile
 * k * world_size + rank).  Asynchronous on the context's stream; pixels of edge tiles that fall
 * outside the image are written as 0.  `stats` may be NULL; when given, the call synchronises. */
int rtk_render_tiles_device(rtk_context* ctx, const rtk_camera* cam, const rtk_params* p, float* d_tile_sums,
                            rtk_stats* stats);

/* The same render for callers that CANNOT ALLOCATE device memory — tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb an allocator (it has not been tried under HIP graph capture: the call records timing events
 * and, on its first use on a device, resolves the array's address).  No counterpart in the reference (it has no device).  Arguments, output layout, bits
 * of the result and asynchrony are those of rtk_render_tiles_device; the difference is where the kernels' working
 * memory lives: pixel accumulators, packet queue, counters and the run table sit in ONE static device array of the
 * library (24 MiB + 8.5 KiB per GPU, held until the process ends) instead of context buffers grown with hipMalloc.
 * The call makes no hipMalloc / hipFree and no allocating library call (rtk_set_scene, which does allocate, must have
 * run before; pass stats = NULL to also avoid the 32-byte counter read-back and its synchronisation).
 * Limits: at most RTK_NOALLOC_MAX_TILES local tiles (1 048 576 pixels of this rank; 1200 x 800 fits) and 1024 sample
 * runs per pixel -> RTK_ERR_TOO_LARGE beyond.  Every call on a GPU shares that GPU's array: the call clears what it
 * needs at its start, and CALLS THAT OVERLAP ON ONE GPU (two contexts, two streams) RACE — serialise them.
 * Statistics: `stats`, when given, are those of THIS call alone (frames = 1); the context's own counters — the frames
 * rendered through rtk_render_tiles_device since rtk_reset_stats, read by rtk_get_stats — are neither reset nor
 * advanced by it. */
#define RTK_NOALLOC_MAX_TILES 16384
int rtk_render_tiles_device_noalloc(rtk_context* ctx, const rtk_camera* cam, const rtk_params* p, float* d_tile_sums,
                                    rtk_stats* stats);

/* Counters accumulate over frames rendered with stats == NULL.  rtk_reset_stats zeroes them (asynchronous);
 * rtk_get_stats synchronises the stream and reads them, with the render kernel's HIP-event time summed
 * over the frames since the reset.  (A render call that is GIVEN a stats pointer resets, renders, reads.) */
int rtk_reset_stats(rtk_context* ctx);
int rtk_get_stats(rtk_context* ctx, rtk_stats* stats);

/* Scatter gathered tile buffers back to scanline order (row 0 = top image row, i.e. y = H-1, the
 * order main.cpp:64 emits): `d_gathered` is [world_size][tiles_per_rank_padded][64][3] floats in
 * DEVICE memory, `d_image` is [height][width][3] floats in DEVICE memory.  Asynchronous. */
int rtk_deinterleave_device(rtk_context* ctx, const float* d_gathered, int64_t tiles_per_rank_padded,
                            const rtk_params* p, float* d_image);

/* The same for quantised bytes (rank-local tiles passed through rtk_quantise_device before the gather: a
 * quarter of the bytes on xGMI): `d_gathered` is [world_size][tiles_per_rank_padded][64][3] bytes, `d_image`
 * [height][width][3] bytes = the pixel payload of the reference's image_binary.ppm (main.cpp:57-76).  SURVEY.md §8f-1. */
int rtk_deinterleave_u8_device(rtk_context* ctx, const uint8_t* d_gathered, int64_t tiles_per_rank_padded,
                               const rtk_params* p, uint8_t* d_image);

/* Whole image on one device into HOST memory: out_rgb_sum is [height][width][3] fp32 colour sums,
 * row 0 = top row.  Equivalent to the reference's loops main.cpp:64-76 with write_color_binary's
 * input captured instead of written.  p->rank/world_size must be 0/1 (or 0/0). */
int rtk_render(rtk_context* ctx, const rtk_camera* cam, const rtk_params* p, float* out_rgb_sum, rtk_stats* stats);

/* write_color_binary (Color.hpp:28-48) for n_pixels RGB sums: sqrt(sum/spp), clamp [0,1],
 * (uint8_t)(255*x).  DEVICE pointers, asynchronous.  Byte-exact with the reference for equal sums. */
int rtk_quantise_device(rtk_context* ctx, const float* d_rgb_sum, int64_t n_pixels, int32_t spp, uint8_t* d_rgb_u8);

/* rtk_render followed by write_color_binary on the device: out_rgb_u8 is [height][width][3] bytes in HOST memory,
 * exactly the bytes main.cpp:64-76 streams after the PPM header (Color.hpp:28-48), a quarter of rtk_render's
 * device-to-host copy.  p->sample_offset must be 0 (a complete frame of p->spp samples). */
int rtk_render_u8(rtk_context* ctx, const rtk_camera* cam, const rtk_params* p, uint8_t* out_rgb_u8, rtk_stats* stats);

/* HittableList::hit (HittableList.hpp:30-42) with Sphere3::hit (Sphere3.hpp:31-59) for a batch of
 * rays against the context's scene: HOST arrays; rays are 6 floats (origin, direction; direction
 * NOT normalised, Ray3.hpp:14).  out_hit[i] in {0,1}; out_index[i] = list index of the winning
 * sphere or -1; out_t[i] = hit_record::t or 0.  hit and index are bit-exact with the reference,
 * t is computed with the reference's operation sequence (no FMA contraction, IEEE sqrt and divide). */
int rtk_intersect_batch(rtk_context* ctx, const float* rays, size_t n, float t_min, float t_max, uint8_t* out_hit,
                        int32_t* out_index, float* out_t);

/* Same, DEVICE pointers, asynchronous on the context's stream (used by bench/profiling). */
int rtk_intersect_batch_device(rtk_context* ctx, const float* d_rays, size_t n, float t_min, float t_max,
                               uint8_t* d_hit, int32_t* d_index, float* d_t);

/* hit_record construction (Sphere3.hpp:51-56, Hittable.hpp:15-18) + Material::scatter
 * (Material.cpp:6-54) for a batch: ray i hit sphere index[i] at t[i].  The random inputs are supplied
 * by the caller as 4 uniforms in [0,1) per ray (in rendering they come from the counter-based RNG):
 * Lambertian maps u[0..1] to a unit vector, Metal maps u[0..2] to a point in the unit ball, Glass
 * compares the Schlick reflectance with u[3] (evaluated unconditionally; the reference's lazy draw at
 * Material.cpp:40 is an RNG-stream detail).  HOST arrays.  out7 = scattered origin, direction,
 * and 1.0/0.0 for scatter's return value. */
int rtk_scatter_batch(rtk_context* ctx, const float* rays, const int32_t* index, const float* t, const float* u4,
                      size_t n, float* out7);

/* Camera::get_ray (Camera.hpp:25-29) + the jitter of main.cpp:69-70 exactly as the render kernel
 * evaluates them for (pixel x, row-from-top, sample): HOST output of n rays (6 floats) for the
 * given integer triples xyz[3*i..]; for tests. */
int rtk_primary_rays(rtk_context* ctx, const rtk_camera* cam, const rtk_params* p, const int32_t* xys, size_t n,
                     float* out_rays);

/* The counter-based generator itself (Philox4x32-10): out4[4*i..] for counters ctr4[4*i..] under
 * p->seed; HOST arrays; for known-answer tests. */
int rtk_philox_batch(rtk_context* ctx, uint64_t seed, const uint32_t* ctr4, size_t n, uint32_t* out4);

/* The direct sampling maps the kernels use in place of the reference's rejection samplers (same distributions):
 * kind 0 = uniform in the unit ball (Vector3.cpp:3-9), 1 = uniform on the unit sphere (Vector3.cpp:49-51),
 * 2 = uniform in the unit disk (Vector3.cpp:53-59; z = 0).  u4 = n x 4 uniforms in [0,1) as the kernels hand them
 * over (ball: u0 u1 u2; sphere: u0 u1; disk: u2 u3 — the slots the render kernel uses); out3 = n x 3.  Parity entry point. */
int rtk_sample_batch(rtk_context* ctx, int32_t kind, const float* u4, size_t n, float* out3);

/* Block until everything queued on the context's stream has finished. */
int rtk_synchronize(rtk_context* ctx);

/* ---- One process, several GPUs ---------------------------------------------------------------------------------
 * The C/C++ caller's way to use a whole node: what the 17-line loop nest of main.cpp:62-78 becomes when the image is
 * sharded by interleaved tiles (rank r = devices[r] renders tiles i % n == r; one context and stream per device; the
 * scene is replicated) and the framebuffer is gathered on devices[0] — RCCL send/recv in one group over communicators
 * from ncclCommInitAll when the devices are distinct (librccl.so.1 is opened on demand; nothing is linked), device
 * copies ordered by events when a device is listed more than once (several ranks on one GPU: a test arrangement) —
 * de-interleaved there and copied to the host.  The image is bit-identical for every n (global-pixel RNG counters,
 * integer sample sums).  No counterpart in the reference (single-threaded, no device).
 * rtk_multi_render: p->rank / p->world_size are ignored; out_rgb_sum ([height][width][3] colour sums) and out_u8 (the
 * write_color_binary bytes, Color.hpp:28-48) are HOST buffers, either may be NULL; per_device (n entries) may be NULL.
 * (Python drives the same sharding with one process per GPU and torch.distributed: raytracinginoneweekend_amd/distributed.py.) */
typedef struct rtk_multi rtk_multi;
int rtk_multi_create(const int32_t* devices, int32_t n_devices, rtk_multi** out);
void rtk_multi_destroy(rtk_multi* m);
int rtk_multi_device_count(const rtk_multi* m);
int rtk_multi_uses_rccl(const rtk_multi* m);
int rtk_multi_set_accel(rtk_multi* m, int32_t mode);
int rtk_multi_set_scene(rtk_multi* m, const rtk_scene* scene);
int rtk_multi_render(rtk_multi* m, const rtk_camera* cam, const rtk_params* p, float* out_rgb_sum, uint8_t* out_u8,
                     rtk_stats* per_device);

#ifdef __cplusplus
}
#endif
#endif /* RTK_H */
