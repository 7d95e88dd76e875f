/*
 * nerfhip.h -- C ABI of the MI355X-native NeRF render hot path.
 *
 * This is the drop-in boundary.  The reference
 * (metaverse3d2022/Nerf-Cuda) has no FFI layer: its boundary is the C++ class
 * ngp::NerfRender (include/nerf-cuda/nerf_render.h:29-50) calling CUDA kernels
 * and tiny-cuda-nn objects directly.  Every entry point below names the
 * reference interface it replaces; the C++ class that mirrors ngp::NerfRender
 * on top of this header lives in nerf-cuda_amd/host/nerf_render.h.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures: device memory is passed as
 *     void* (a HIP device pointer), streams as void* (hipStream_t, NULL = the
 *     context's own stream).
 *   - every function returns an int status: 0 = NRF_OK, otherwise an
 *     NRF_E_* code; nrf_last_error() returns a thread-local message.
 *     No C++ exception crosses this boundary.
 *   - one context per device; a context is used by one thread at a time.  Render calls of one context may overlap
 *     on different streams: every call takes its own statistics counters and work queues from a ring of 16, so at
 *     most 16 render calls of a context may be in flight; nrf_get_stats reports the last call.
 *   - there is NO CPU fallback: if the HIP runtime finds no gfx950 device,
 *     nrf_create fails with NRF_E_NODEVICE.
 */
#ifndef NERFHIP_H_
#define NERFHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRF_ABI_VERSION 6
#define NRF_MAX_VIEWS 128 /* cameras one launch of the fused kernel takes (nrf_render_views) */

/* ---- status codes ------------------------------------------------------ */
enum {
  NRF_OK = 0,
  NRF_E_INVALID = 1,     /* bad argument / null pointer / bad shape          */
  NRF_E_UNSUPPORTED = 2, /* config the HIP path does not implement           */
  NRF_E_NODEVICE = 3,    /* no usable gfx950 device                          */
  NRF_E_HIP = 4,         /* a HIP runtime call failed                        */
  NRF_E_STATE = 5,       /* call order violated (e.g. render before load)    */
  NRF_E_PARAMS = 6       /* parameter count / density grid size mismatch:
                            reference nerf_network.h:425-427,
                            nerf_render.cu:467-469                           */
};

/* ---- enums mirrored from the reference's JSON vocabulary ---------------- */
/* tcnn activation names, T/src/network.cu:41-60 */
enum {
  NRF_ACT_NONE = 0,
  NRF_ACT_RELU = 1,
  NRF_ACT_EXPONENTIAL = 2,
  NRF_ACT_SIGMOID = 3,
  NRF_ACT_SQUAREPLUS = 4,
  NRF_ACT_SOFTPLUS = 5,
  NRF_ACT_SINE = 6
};
/* direction encodings, T/src/encoding.cu:97-117 */
enum {
  NRF_DIR_SH = 0,        /* SphericalHarmonics, degree 1..8                     */
  NRF_DIR_FREQUENCY = 1, /* Frequency, n_frequencies 1..18 (tcnn's default: 12)  */
  NRF_DIR_IDENTITY = 2
};
/* hash-grid flavours, T/include/tiny-cuda-nn/encodings/grid.h:1366-1367 */
enum { NRF_GRID_HASH = 0, NRF_GRID_DENSE = 1, NRF_GRID_TILED = 2 };
/* grid interpolation, T/include/tiny-cuda-nn/encodings/grid.h:1383 ("interpolation": Nearest | Linear | Smoothstep),
 * kernel_grid :196-232 */
enum { NRF_INTERP_LINEAR = 0, NRF_INTERP_NEAREST = 1, NRF_INTERP_SMOOTHSTEP = 2 };

/* ---- model description ---------------------------------------------------
 * Replaces: NerfRender::load_snapshot + reset_network + NerfNetwork ctor +
 * NerfNetwork::deserialize (nerf_render.cu:431-473,111-184;
 * nerf_network.h:95-144,424-443).  The caller has already parsed the
 * msgpack snapshot; all pointers are HOST pointers and are copied.           */
typedef struct nrf_model_desc {
  uint32_t abi_version; /* = NRF_ABI_VERSION */

  /* "encoding" block (position hash grid, 3-D input) */
  uint32_t grid_type;             /* NRF_GRID_*                               */
  uint32_t n_levels;              /* L, <= 16                                 */
  uint32_t n_features_per_level;  /* F in {1, 2, 4, 8} (grid.h:1403-1411)     */
  uint32_t log2_hashmap_size;     /* log2 T                                   */
  uint32_t base_resolution;       /* Nmin                                     */
  float per_level_scale;          /* b (already derived, nerf_render.cu:158)  */
  uint32_t interpolation;         /* NRF_INTERP_*                             */

  /* "network" (density MLP) and "rgb_network" blocks: FullyFusedMLP widths 16/32/64/128, any number of hidden
   * layers >= 1 (T/src/fully_fused_mlp.cu:636-687, 700-725).  L = 16, F = 2, 64 neurons, 1 + 2 hidden layers, a
   * 16-wide direction encoding, ReLU / None / Exponential activations (the reference's base.json) run in the
   * register-resident instance of the fused kernel; every other combination in its generic instance.           */
  uint32_t n_neurons;
  uint32_t density_hidden_layers;  /* n_hidden_layers of "network"            */
  uint32_t density_activation;     /* hidden activation                       */
  uint32_t density_output_activation;
  uint32_t density_n_output;       /* 1..16, padded to 16 rows (nerf_network.h:120-122) */
  uint32_t sigma_activation;       /* nerf_network.h:125, default Exponential */
  uint32_t rgb_hidden_layers;
  uint32_t rgb_activation;
  uint32_t rgb_output_activation;

  /* "dir_encoding" block (Composite{SH|Frequency|Identity on 3 dims}) */
  uint32_t dir_encoding;  /* NRF_DIR_*                                        */
  uint32_t sh_degree;     /* for NRF_DIR_SH                                   */
  uint32_t n_frequencies; /* for NRF_DIR_FREQUENCY                            */

  /* "snapshot" block, nerf_render.cu:441-453 */
  float aabb[6];
  float bound;
  float scale;
  uint32_t cascade;           /* C */
  uint32_t density_grid_size; /* H */
  float mean_density;

  /* fp32 parameters in the reference order (nerf_network.h:273-291):
   * density MLP | rgb MLP | hash grid | (dir encoding: none).
   * Each MLP: first [n_neurons x in] | hidden [n_neurons x n_neurons]... |
   * last [16 x n_neurons], row-major [out][in], no biases.                   */
  const float* params;
  uint64_t n_params;
  /* float density grid [C*H*H*H], index level*H^3 + x*H^2 + y*H + z; NULL (n = 0): the snapshot carries
   * none -- nrf_generate_density_grid evaluates one from the network before the first render              */
  const float* density_grid;
  uint64_t n_density_grid;
  /* (ABI 6) Device memory the library may spend on GATHER COPIES of the hash grid, in MB; 0 = the default: a sixteenth of the
   * device's memory (MI355X: 18 GB), at most half of what is free; 1 = none.  For the base.json grid shape (L = 16, F = 2, Linear) the render kernel
   * reads a level's eight trilinear corners as two aligned 16-byte "quads" from a cell-major copy of the level (every entry
   * copied, on the device, from the index grid.h:100-117 names: the same bits) instead of eight 4-byte table entries -- a
   * quarter of the gather addresses.  Copies are made four levels at a time, in level order, while they fit: levels 0..7 of
   * base.json's grid take 95 MB, levels 8..11 another 4.5 GB (MI355X, 1080p: 11.7 -> 10.7 -> 10.1 ms per 16 views).  The table
   * the reference defines is kept beside them.  The environment variable NRF_QUAD_BUDGET_MB overrides this field.         */
  uint32_t gather_copy_budget_mb;
} nrf_model_desc;

/* Level geometry derived on the host exactly as the reference does
 * (grid.h:899-931 ctor, grid.h:186-190 kernel).                              */
typedef struct nrf_level_table {
  uint32_t n_levels;
  uint32_t offset[17];     /* in entries (F values each); offset[L] = total   */
  uint32_t resolution[16]; /* grid_resolution = ceil(scale)+1                 */
  float scale[16];         /* exp2f(l*log2f(b))*Nmin - 1                      */
} nrf_level_table;

/* Render parameters.  The reference keeps these as private members without
 * setters (nerf_render.h:55-78); defaults here equal those members.          */
typedef struct nrf_options {
  float bg_color;      /* 1      m_bg_color (int in the reference)            */
  float min_near;      /* 0.2    m_min_near                                   */
  float dt_gamma;      /* 1/128  m_dt_gamma                                   */
  int32_t max_steps;   /* 1024   m_max_infer_steps                            */
  float density_scale; /* 1      m_density_scale                              */
  int32_t perturb;     /* 0      m_perturb.  > 0: the seed of kernel_march_rays' perturb branch (render_utils.h:550,
                          585-589: t += MIN_STEPSIZE() * pcg32(n, perturb).next_float() ahead of every march call) -- in
                          nrf_march n = the ray's place in the call; in rendered frames n = the ray's pixel and every sample's
                          search is a call (the per-ray loop), in instances of the per-strip kernel (slower than the default
                          path: the reference never takes this branch, m_perturb = false).  < 0: NRF_E_INVALID            */
  /* Sharding of one frame over ranks: the frame is cut into 8x8-pixel tiles
   * and those into strips of 4 horizontally adjacent tiles (32x8 pixels, one
   * workgroup); strip id = ty*ceil(ceil(W/8)/4) + tx/4, and strip id %
   * shard_count == shard_index belongs to this context, as local strip
   * id / shard_count (local tile = 4*local strip + tx%4).  Replaces the pixel
   * interleave p = NGPU*tid + gpu of render_utils.h:37.                      */
  int32_t shard_index; /* 0 */
  int32_t shard_count; /* 1 */
  /* Opt-in (default 0), NOT the reference's arithmetic: the trilinear interpolation of the hash grid accumulates
   * (half)(w * h + acc) with one rounding per corner (v_fma_mixlo/hi_f16) instead of the reference's three (grid.h:258-260:
   * fp32 product, cast to fp16, fp16 sum) -- half the vector instructions of the kernel's most expensive loop.  Features
   * differ from the bit-exact path by an ulp of a partial sum now and then (more accurate, not identical: within 4 x 2^-11
   * for table entries of magnitude 0.5); frames within the 2/255 the parity tests allow.  Measured gain: under 1 % -- the
   * kernel is co-limited by its gather path (DESIGN.md "What binds the kernel").  Honoured by the register-resident instance of the fused kernel in its persistent form and by
   * nrf_encode_grid; other instances render with the exact arithmetic.                                              */
  int32_t fast_interp; /* 0 */
  /* 1: a single shard (shard_count == 1) is rendered in the shard layout as well -- tile-major [n_tiles][64], what
   * nrf_untile* takes -- instead of the row-major frame.  What a one-rank rehearsal of the multi-GPU exchange renders
   * (bench.py --force-dist: render -> RCCL gather -> untile with shard_count 1); host frames and nrf_read_* need 0.  */
  int32_t tile_major;  /* 0 */
} nrf_options;

/* One rendered frame (device memory owned by the context, valid until the
 * next nrf_render on the same context).  Replaces ngp::Image
 * (common.h:75-89) plus the float buffers image/depth/weight_sum
 * (nerf_render.cu:200-202).                                                  */
typedef struct nrf_frame {
  int32_t width, height;
  int32_t n_tiles;      /* tiles rendered by this shard (4 per strip, including
                           out-of-image padding tiles of a ragged strip)      */
  void* rgba;           /* device float [n_px][4]; rgb after background blend
                           (render_utils.h:259-261), a = weight_sum           */
  void* depth;          /* device float [n_px] (render_utils.h:262-263)       */
  /* layout: shard_count==1 (and nrf_options.tile_major == 0) -> row-major
   * [H][W]; otherwise tile-major [n_tiles][64] in ascending tile id, see
   * nrf_untile.                                                              */
  int32_t tile_major;
  /* batched renders (nrf_render_views): view i lives at rgba + i*view_stride_px
   * pixels (depth likewise); n_views == 1 for nrf_render                      */
  int32_t n_views;
  int64_t view_stride_px;
} nrf_frame;

typedef struct nrf_stats {
  uint64_t n_rays;      /* rays generated by this shard                       */
  uint64_t n_samples;   /* march-emitted samples evaluated by the network     */
  uint64_t n_rounds;    /* sum over wave tiles of march/eval/composite rounds */
  uint64_t n_network_evals; /* network evaluations including the padding of the
                           16-sample MFMA tiles (>= n_samples)                 */
  float render_ms;      /* device time of the last nrf_render (hipEvents)     */
  uint64_t n_composited; /* samples that entered a ray's compositing sum: what the reference's own per-ray schedule emits
                           (n_samples also counts the samples a ray queues behind its terminating one; how many those are
                           depends on how rays are batched into rounds, this count does not)                           */
  float shader_clock_mhz; /* the core clock the last launch of the persistent render kernel ran at, measured in the launch:
                           d(s_memtime) / d(s_memrealtime) x 100 MHz between the entry and the exit of one wave per workgroup,
                           summed over the workgroups; 0 when the last render did not run that kernel                    */
  uint32_t gather_addresses_per_sample; /* (ABI 6) lane addresses one evaluated sample sends into the texture path with the loaded
                           model: 8 per level (the corners of grid.h:236-262), 2 for a level that is gathered from its cell-major quad
                           copy (two aligned 16-byte entries), 1 per level with Nearest interpolation                              */
  uint64_t grid_device_bytes; /* (ABI 6) device memory of the loaded model's hash grid: the reference-order table + its gather copies
                           (nrf_model_desc.gather_copy_budget_mb)                                                                  */
} nrf_stats;

typedef struct nrf_context nrf_context;

/* ---- lifecycle ----------------------------------------------------------- */
const char* nrf_last_error(void);
int nrf_abi_version(void);
/* NerfRender::NerfRender(), nerf_render.cu:46-57 (per-GPU stream creation)   */
int nrf_create(int device, nrf_context** out);
int nrf_destroy(nrf_context* ctx);
void nrf_default_options(nrf_options* o);

/* host-only helper: level geometry + parameter count for a description.
 * n_params check of nerf_network.h:425.                                      */
int nrf_level_table_compute(const nrf_model_desc* d, nrf_level_table* t);
int nrf_expected_n_params(const nrf_model_desc* d, uint64_t* n);
/* host-only: the reference's automatic per_level_scale, nerf_render.cu:158-165:
 * fp32 exp(log(2048*bound/base_resolution)/(n_levels-1)).                    */
int nrf_default_per_level_scale(float bound, uint32_t base_resolution, uint32_t n_levels,
                                float* out);

/* load_snapshot + reset_network + deserialize                                */
int nrf_load_model(nrf_context* ctx, const nrf_model_desc* d);
/* NerfRender::generate_density_grid, nerf_render.cu:388-429 (dead and incomplete in the reference: its density
 * query is commented out at :415).  Completed as its origin (torch-ngp's update_extra_state) defines it: for every
 * cascade, the density network at every cell position (init_xyzs / dd_scale, render_utils.h:79-108; no random
 * perturbation), scaled by 0.001691, folded into a grid that starts at 1/64 by g = max(g * decay, value)
 * (dg_update, render_utils.h:120-128) n_iterations times; mean_density = mean(max(g, 0)).  Replaces the model's
 * density grid and rebuilds the occupancy structures of the march.  nrf_read_density_grid returns the float grid
 * [cascade * H^3] the march currently uses (snapshot's or generated) and its mean_density.                        */
int nrf_generate_density_grid(nrf_context* ctx, int n_iterations, float decay, float* mean_density);
int nrf_read_density_grid(nrf_context* ctx, float* grid, uint64_t n, float* mean_density);
/* NerfRender::set_resolution, nerf_render.cu:186-236 (idempotent here)       */
int nrf_set_resolution(nrf_context* ctx, int width, int height);
int nrf_set_options(nrf_context* ctx, const nrf_options* o);

/* ---- the hot path -------------------------------------------------------- */
/* NerfRender::render_frame, nerf_render.cu:238-367.
 * cam = {fl_x, fl_y, cx, cy} (common.h:68-74); pose = row-major 4x4
 * camera-to-world in the NeRF/Blender convention (converted with
 * nerf_matrix_to_ngp, render_utils.h:68-77).  Asynchronous on `stream`
 * unless stream==NULL, in which case the call returns after completion.
 * A camera more than 4096 units (the reference's ngp units: 0.33 x the pose's
 * translation + 0.5) from the origin renders as background: out there t + dt
 * stops changing t in fp32, and the march of render_utils.h:593-653 -- the
 * reference's as well as this one -- would never end.                         */
int nrf_render(nrf_context* ctx, const float cam[4], const float pose[16],
               void* stream, nrf_frame* out);
/* Batched multi-view render: n_views cameras (cams [n][4], poses [n][16], same
 * model / resolution / shard) in ONE launch per NRF_MAX_VIEWS views: the tiles
 * of view v+1 take the wave slots the few long tiles at the end of view v leave
 * idle (the kernel is persistent -- one workgroup per compute unit pulls tile
 * strips of all views from work queues -- so every launch ends with the same
 * short tail whatever its size, and nothing else runs on the device while it
 * is resident: see INTEGRATION.md).  This is the path for a render_server with many concurrent camera
 * requests (the reference serves them one render_frame at a time,
 * render_server.cu:86-101) and for multi-GPU shards, which are too small to
 * fill a GPU alone.  Every view is bit-identical to an nrf_render of that
 * camera.  The context's own buffers hold nrf_set_max_views views (default 1);
 * bound buffers (nrf_bind_output) must hold n_views * view_stride_px pixels.  */
int nrf_set_max_views(nrf_context* ctx, int max_views);
int nrf_render_views(nrf_context* ctx, int n_views, const float* cams, const float* poses,
                     void* stream, nrf_frame* out);
/* SURVEY.md 8(b) lists this entry point as nrf_render_batch: same function.   */
int nrf_render_batch(nrf_context* ctx, int n_views, const float* cams, const float* poses,
                     void* stream, nrf_frame* out);
/* Binds caller-owned device buffers (e.g. a render buffer's RGBA plane or a
 * torch tensor) as the target of subsequent nrf_render calls: rgba float
 * [n_px][4], depth float [n_px], n_px as nrf_frame describes.  NULL, NULL
 * returns to the context's own buffers.  Replaces the host round trip
 * render_frame -> host_to_accumulate_buffer of main.cu:87-129.               */
int nrf_bind_output(nrf_context* ctx, void* rgba, void* depth);
/* Binds a caller-owned device buffer of packed 8-bit pixels -- the reference's output format
 * (unsigned char)(255.0 * x) of r, g, b and depth (nerf_render.cu:352-359) as r | g << 8 | b << 16 | depth << 24,
 * uint32 [n_views * view_stride_px], same pixel order as the float planes (tile-major for a shard) -- as the
 * target of subsequent renders: the kernel writes these 4 bytes per pixel instead of the 20 of the float planes,
 * bit-identical to nrf_quantize_rgbd8 of them.  The frame is then the caller's to read (nrf_read_* return
 * NRF_E_STATE, nrf_frame::rgba / depth are NULL); NULL returns to the float planes.  What a rank of a multi-GPU
 * step binds: its shard goes onto the wire as rendered.                                                    */
int nrf_bind_output_rgbd8(nrf_context* ctx, void* rgbd8);
/* Binds caller-owned device (or device-visible pinned host) buffers in the layout of the reference's host Image
 * (common.h:75-89, filled by nerf_render.cu:352-359): rgb u8 [n_views * view_stride_px][3], depth u8
 * [n_views * view_stride_px], row-major, single-shard frames only; both 4-byte aligned.  The kernel writes these
 * 4 bytes per pixel itself -- bit-identical to nrf_read_u8 of the float planes -- as whole dwords assembled inside
 * the tile's wavefront.  nrf_read_* return NRF_E_STATE; NULL, NULL returns to the float planes.               */
int nrf_bind_output_u8(nrf_context* ctx, void* rgb8, void* depth8);
/* nrf_render on the context's own stream WITHOUT waiting for it (the way the
 * reference overlaps its NGPU devices, nerf_render.cu:252-362); nrf_sync waits. */
int nrf_render_async(nrf_context* ctx, const float cam[4], const float pose[16], nrf_frame* out);
int nrf_sync(nrf_context* ctx);
/* ---- host frames: NerfRender::render_frame's result is HOST memory (nerf_render.cu:345-359: D2H of the float planes and
 * a single-threaded quantise / de-interleave loop per GPU; Image, common.h:75-89).  nrf_submit_host_u8 renders n_views
 * cameras into 8-bit planes (the kernel writes the Image's bytes itself), has the copy engine move the rows of every
 * view's region of interest into pinned host memory owned by the context, and fills the remaining rows -- background by
 * construction -- from the calling thread while the GPU renders.  It returns at once; nrf_wait_host_u8 blocks until the
 * frames of that ticket are in host memory.  Two slots: the copy of one call overlaps the render of the next, and the
 * pointers of a ticket stay valid until the SECOND next nrf_submit_host_u8 on the context (that call waits for the
 * slot's copy, not for its reader).  NRF_HOST_RGB_ONLY: depth is not copied (nrf_host_frame::depth = NULL) -- what a
 * render_server needs.  Bytes identical to nrf_render + nrf_read_u8.  Single-shard frames only.                      */
enum { NRF_HOST_RGB_ONLY = 1 };
typedef struct nrf_host_frame {
  int32_t width, height, n_views;
  const uint8_t* rgb;      /* pinned host: view v at rgb + v * view_stride_px * 3, [H][W][3] */
  const uint8_t* depth;    /* pinned host: view v at depth + v * view_stride_px, [H][W]; NULL with NRF_HOST_RGB_ONLY */
  int64_t view_stride_px;
  float render_ms;         /* device time of this call's render launches (events on their stream) */
  uint64_t copied_bytes;   /* what the copy engine moved for this call (the rows of the regions of interest) */
} nrf_host_frame;
int nrf_submit_host_u8(nrf_context* ctx, int n_views, const float* cams, const float* poses, int flags, int* ticket);
int nrf_wait_host_u8(nrf_context* ctx, int ticket, nrf_host_frame* out);
/* submit + wait: one NerfRender::render_frame(s) call of the reference                                               */
int nrf_render_host_u8(nrf_context* ctx, int n_views, const float* cams, const float* poses, int flags, nrf_host_frame* out);
/* Host copy + quantisation, nerf_render.cu:345-359 (saturating, row-major), of a frame rendered into float planes
 * (a quantise launch + two blocking copies per view: the after-the-fact path; nrf_render_host_u8 is the fast one). */
int nrf_read_u8(nrf_context* ctx, uint8_t* rgb, uint8_t* depth);
/* The same quantisation on the device for any float frame / shard / batch of
 * n_px pixels: out[i] = r | g << 8 | b << 16 | depth << 24.  Four bytes per
 * pixel instead of twenty is what a multi-GPU gather should carry when the
 * consumer wants the reference's 8-bit Image (common.h:75-89); 4-byte pixels go
 * through nrf_untile(_views) with channels = 1.                               */
int nrf_quantize_rgbd8(nrf_context* ctx, const void* rgba, const void* depth, uint64_t n_px,
                       void* out_u32, void* stream);
/* The same quantisation into the planar layout of the reference's Image: rgb u8 [n_px][3], depth u8 [n_px] (device). */
int nrf_quantize_u8(nrf_context* ctx, const void* rgba, const void* depth, uint64_t n_px, void* rgb8, void* depth8,
                    void* stream);
/* Host copy of the float buffers (row-major; single-shard frames only).      */
int nrf_read_f32(nrf_context* ctx, float* rgba, float* depth);
/* nrf_read_f32 / nrf_read_u8 for view `view` of the last nrf_render_views.    */
int nrf_read_view_f32(nrf_context* ctx, int view, float* rgba, float* depth);
int nrf_read_view_u8(nrf_context* ctx, int view, uint8_t* rgb, uint8_t* depth);
/* Host copy of this context's shard as rendered: rgba [n_tiles*64][4], depth
 * [n_tiles*64] (tile-major when shard_count > 1, else the row-major frame).  */
int nrf_read_shard_f32(nrf_context* ctx, float* rgba, float* depth);
/* Rearranges gathered tile-major shards ([shard][n_tiles_max][64][C] floats,
 * as produced by a gather of every rank's nrf_frame buffer) into a
 * row-major [H][W][C] image on the device.  Replaces the de-interleave loop
 * nerf_render.cu:352-359.                                                    */
int nrf_untile(nrf_context* ctx, const void* gathered, int shard_count,
               int tiles_per_shard, int channels, void* out_rowmajor,
               void* stream);
/* Same for gathered batches of nrf_render_views: [shard][view][n_tiles_max][64][C]
 * -> [view][H][W][C].                                                         */
int nrf_untile_views(nrf_context* ctx, const void* gathered, int shard_count,
                     int tiles_per_shard, int channels, int n_views,
                     void* out_rowmajor, void* stream);
/* Gathered PACKED shards ([shard][view][n_tiles_max][64] uint32 r | g << 8 | b << 16 | depth << 24, as rendered with
 * nrf_bind_output_rgbd8) -> the planar layout of the reference's Image: rgb u8 [view][H][W][3], depth u8 [view][H][W]
 * (device or device-visible pinned host memory; the shards 16-byte aligned, the planes 4-byte aligned when the width is a multiple of 4).  The de-interleave + quantise loop of
 * nerf_render.cu:352-359 for a device group, on the device.                                                    */
int nrf_untile_views_u8(nrf_context* ctx, const void* gathered_rgbd8, int shard_count, int tiles_per_shard, int n_views,
                        void* rgb8, void* depth8, void* stream);
int nrf_tiles_per_shard(int width, int height, int shard_count, int* n);
int nrf_get_stats(nrf_context* ctx, nrf_stats* s);

/* ---- device groups: several GPUs driven by one process ---------------------
 * The reference's NGPU threads + D2H gather + host de-interleave loop
 * (nerf_render.cu:252-362, common.h:91) as one object: every member renders
 * its strips of every view, ships the shard device-to-device to devices[0]
 * (xGMI peer copies) and devices[0] untiles.  devices == NULL means 0..n-1; a
 * device may be listed more than once.  The frames of nrf_group_render_views
 * are row-major float planes on devices[0] (nrf_frame, view_stride_px = W*H),
 * bit-identical to nrf_render_views on a single context.  The one-process-per-
 * GPU form (torch.distributed / RCCL gather) is bench.py.                      */
typedef struct nrf_group nrf_group;
int nrf_group_create(int n_devices, const int* devices, nrf_group** out);
int nrf_group_destroy(nrf_group* grp);
/* The transport of the one exchange step (the reference: cudaMemcpyAsync D2H of every GPU's planes into one host buffer,
 * nerf_render.cu:345-359).  NRF_GATHER_PEER_COPY (default): hipMemcpyPeerAsync per member.  NRF_GATHER_RCCL: one RCCL
 * communicator per member (ncclCommInitAll) and one group of ncclSend / ncclRecv to devices[0] per call -- a direct gather
 * over the point-to-point xGMI links; needs distinct devices (NRF_E_UNSUPPORTED otherwise) and librccl, which is opened
 * on this call (a process that never asks never loads it).  In this mode a ONE-member group runs the whole exchange too
 * (tile-major shard, send to self, untile).  Frames are the same bits either way.  May be called between renders; it
 * reallocates the group's buffers, so host-frame pointers handed out earlier become invalid and a ticket that has not been
 * waited for makes it return NRF_E_STATE.  The environment variable NRF_GROUP_GATHER=rccl|peer sets it at nrf_group_create. */
enum { NRF_GATHER_PEER_COPY = 0, NRF_GATHER_RCCL = 1 };
int nrf_group_set_gather(nrf_group* grp, int mode);
/* rccl_version: ncclGetVersion() of the opened library in RCCL mode (e.g. 22703), else 0 */
int nrf_group_get_gather(const nrf_group* grp, int* mode, int* rccl_version);
int nrf_group_size(const nrf_group* grp);
/* member i's context (owned by the group): stage entry points, per-device statistics */
nrf_context* nrf_group_member(nrf_group* grp, int index);
int nrf_group_load_model(nrf_group* grp, const nrf_model_desc* d);
int nrf_group_set_options(nrf_group* grp, const nrf_options* o); /* shard_* are set by the group */
int nrf_group_set_resolution(nrf_group* grp, int width, int height);
int nrf_group_render_views(nrf_group* grp, int n_views, const float* cams, const float* poses,
                           nrf_frame* out);
int nrf_group_read_view_f32(nrf_group* grp, int view, float* rgba, float* depth);
int nrf_group_read_view_u8(nrf_group* grp, int view, uint8_t* rgb, uint8_t* depth);
/* Host frames of a group (see nrf_submit_host_u8): the members render packed 8-bit shards, devices[0] untiles them
 * into the Image layout and ONE copy per call brings all views to pinned host memory; same ticket rules.          */
int nrf_group_submit_host_u8(nrf_group* grp, int n_views, const float* cams, const float* poses, int flags, int* ticket);
int nrf_group_wait_host_u8(nrf_group* grp, int ticket, nrf_host_frame* out);
int nrf_group_render_host_u8(nrf_group* grp, int n_views, const float* cams, const float* poses, int flags, nrf_host_frame* out);
int nrf_group_get_stats(nrf_group* grp, nrf_stats* s); /* sums; render_ms = slowest member */

/* ---- stage entry points (unit parity against the oracle) -----------------
 * All pointers are DEVICE pointers, n = number of samples / rays.            */
/* kernel_grid<half,3,F>, grid.h:139-268.  pos01 [n][3] in [0,1];
 * out fp16 [n][next_multiple(L*F, 16)] (padding columns are 0, grid.h:959-969) */
int nrf_encode_grid(nrf_context* ctx, const void* pos01, uint32_t n, void* out_f16,
                    void* stream);
/* kernel_sh / frequency_encoding / identity.  dir01 [n][3] in [0,1];
 * out fp16 [n][next_multiple(raw width, 16)]: 16 for SH <= 4, 64 for SH 8,
 * 80 for Frequency with 12 frequencies                                       */
int nrf_encode_dir(nrf_context* ctx, const void* dir01, uint32_t n, void* out_f16,
                   void* stream);
/* kernel_mlp_fused x2 + extract_density (nerf_network.h:148-196) on
 * pre-encoded inputs: feat fp16 [n][feature width], dirfeat fp16 [n][dir
 * width] (the widths of nrf_encode_grid / nrf_encode_dir: 32 and 16 for the
 * base configuration) -> out fp16 [n][4] = (r,g,b,sigma)                     */
int nrf_mlp_forward(nrf_context* ctx, const void* feat_f16, const void* dirfeat_f16,
                    uint32_t n, void* out_f16, void* stream);
/* Measurement aid: the same call, every chunk of samples evaluated `repeat`
 * times from registers (identical output): repeat >> 1 gives the rate of the
 * MFMA chain with its fp32 -> fp16 re-packing, without the HBM stream.       */
int nrf_mlp_forward_repeat(nrf_context* ctx, const void* feat_f16, const void* dirfeat_f16,
                           uint32_t n, void* out_f16, uint32_t repeat, void* stream);
/* Whole network on raw march output (world-space xyz in [-bound,bound], unit
 * dirs), including the two affine maps nerf_render.cu:311-314 and
 * decompose (render_utils.h:308-334): -> sigma f32 [n], rgb f32 [n][3]       */
int nrf_network(nrf_context* ctx, const void* xyz, const void* dir, uint32_t n,
                void* sigma_f32, void* rgb_f32, void* stream);
/* set_rays_o/d + kernel_near_far_from_aabb for every pixel (row-major):
 * rays_o [n][3], rays_d [n][3], nears [n], fars [n]                          */
int nrf_generate_rays(nrf_context* ctx, const float cam[4], const float pose[16],
                      void* rays_o, void* rays_d, void* nears, void* fars,
                      void* stream);
/* Same with HOST output arrays (any may be NULL); NerfRender::generate_rays
 * (nerf_render.cu:369-386) for callers that hold no device memory.           */
int nrf_generate_rays_host(nrf_context* ctx, const float cam[4], const float pose[16],
                           float* rays_o, float* rays_d, float* nears, float* fars);
/* kernel_march_rays (render_utils.h:524-655) for n rays with explicit start
 * t: xyzs [n][n_step][3], dirs likewise, deltas [n][n_step][2]; unused slots
 * are zero-filled (DESIGN.md deviation D-1).                                 */
int nrf_march(nrf_context* ctx, const void* rays_o, const void* rays_d,
              const void* rays_t, const void* fars, uint32_t n, uint32_t n_step,
              void* xyzs, void* dirs, void* deltas, void* stream);
/* kernel_composite_rays (render_utils.h:658-751): state [n][5] =
 * (weight_sum, depth, r, g, b) updated in place; rays_t in/out               */
int nrf_composite(nrf_context* ctx, const void* sigmas, const void* rgbs,
                  const void* deltas, uint32_t n, uint32_t n_step, void* rays_t,
                  void* state, void* stream);

/* ---- render buffer (presentation chain) -----------------------------------
 * Replaces ngp::CudaRenderBuffer (include/nerf-cuda/render_buffer.h:160-315,
 * src/render_buffer.cu:224-259 accumulate_kernel, :261-342 tonemap,
 * :529-556 tonemap_kernel, :590-627).  The frame buffer is RGBA fp32 + depth
 * on the device: bind it as the render target with
 * nrf_bind_output(ctx, rgba, depth) and the kernel composites straight into
 * it (the reference goes through host u8, main.cu:87-129).  GL / CUDA-surface
 * / DLSS / overlay parts of the reference class are out of scope; the
 * tonemapped result goes to a plain RGBA fp32 "surface" buffer.              */
enum { NRF_CS_LINEAR = 0, NRF_CS_SRGB = 1, NRF_CS_VISPOSNEG = 2 };          /* EColorSpace, common.h:93-97     */
enum { NRF_TM_IDENTITY = 0, NRF_TM_ACES = 1, NRF_TM_HABLE = 2, NRF_TM_REINHARD = 3 }; /* ETonemapCurve :100-105 */
typedef struct nrf_render_buffer nrf_render_buffer;
int nrf_rb_create(int device, nrf_render_buffer** out);
int nrf_rb_destroy(nrf_render_buffer* rb);
int nrf_rb_resize(nrf_render_buffer* rb, int width, int height);           /* resize(), zero-fills           */
int nrf_rb_reset_accumulation(nrf_render_buffer* rb);                      /* m_spp = 0                      */
int nrf_rb_spp(nrf_render_buffer* rb, uint32_t* spp);
int nrf_rb_set_color_space(nrf_render_buffer* rb, int color_space);        /* m_color_space                  */
int nrf_rb_set_tonemap_curve(nrf_render_buffer* rb, int curve);            /* m_tonemap_curve                */
/* device pointers: frame RGBA [h][w][4], depth [h][w], accumulate RGBA, surface RGBA (tonemap output) */
int nrf_rb_buffers(nrf_render_buffer* rb, void** frame, void** depth, void** accumulate, void** surface);
int nrf_rb_clear_frame(nrf_render_buffer* rb, void* stream);               /* clear_frame()                  */
int nrf_rb_accumulate(nrf_render_buffer* rb, float exposure, void* stream);/* accumulate(): running mean     */
int nrf_rb_tonemap(nrf_render_buffer* rb, float exposure, const float background_color[4],
                   int output_color_space, void* stream);                  /* tonemap()                      */
/* accumulate() + tonemap() of the reference's per-frame sequence (main.cu:87-129, render_buffer.cu:590-627) as ONE pass over
 * the planes: reads the frame (and the mean so far, unless this is the first sample), writes the new mean and the developed
 * surface -- 64 B per pixel instead of 96 for the two calls; the planes hold the same bits as after nrf_rb_accumulate +
 * nrf_rb_tonemap.  rgba8 (optional, device, uint32 [h][w]): the surface as 8-bit r | g << 8 | b << 16 | a << 24 (saturating,
 * NaN -> 0: the library's 8-bit rule), what a display or an encoder takes; NULL: not written.                       */
int nrf_rb_present(nrf_render_buffer* rb, float exposure, const float background_color[4],
                   int output_color_space, void* rgba8, void* stream);
/* host_to_accumulate_buffer(): rgb u8 [n][3] -> accumulate RGBA = rgb/255, a = 1 (render_buffer.h:231-241) */
/* overlay_depth() (render_buffer.cu:431-477, 690-714): turbo-coloured depth (device float plane of
 * image_width x image_height, e.g. nrf_frame.depth) blended over the surface with weight alpha.     */
int nrf_rb_overlay_depth(nrf_render_buffer* rb, float alpha, const void* depth, float depth_scale,
                         int image_width, int image_height, int fov_axis, float zoom,
                         const float screen_center[2], void* stream);
int nrf_rb_host_to_accumulate_buffer(nrf_render_buffer* rb, const uint8_t* rgb, int n);
int nrf_rb_read(nrf_render_buffer* rb, float* accumulate_rgba, float* surface_rgba);  /* host copies (either may be NULL) */

#ifdef __cplusplus
}
#endif
#endif /* NERFHIP_H_ */
