// nrf_group.hip -- several GPUs driven by ONE process behind the C ABI (include/nerfhip.h "device groups").
//
// The reference renders on NGPU devices from one process: a std::thread per device, pixel-interleaved
// shards, a D2H copy of every shard and a single-threaded de-interleave loop on the host
// (R/src/nerf_render.cu:252-362).  Here every member context renders its strips of every view of the
// batch on its own stream, ships the tile-major shard device-to-device to the first member and the
// first member untiles all views in one launch.  No host thread, no host copy.  Two transports for
// the one exchange step (nrf_group_set_gather / NRF_GROUP_GATHER):
//   NRF_GATHER_PEER_COPY  hipMemcpyPeerAsync on the member's stream (xGMI peer-to-peer: point-to-point links, so the
//                         n-1 copies into device 0 run on n-1 different links), an event per member, devices[0] waits
//   NRF_GATHER_RCCL       one communicator per member (ncclCommInitAll: one process, n devices) and ONE group of
//                         ncclSend (member i, on its stream, behind its render) / ncclRecv (devices[0], on its stream,
//                         ahead of the untile) per call: a direct gather -- xGMI is point-to-point, a ring all-gather
//                         would move n times the bytes over every link (BASELINE north_star "RCCL over xGMI
//                         only for the final gather").  librccl is opened with dlopen on first use: a process that
//                         never asks for it never loads it.  A one-member group runs the whole exchange in this mode
//                         (tile-major shard, send-to-self, untile): what a one-GPU box can execute of it.
// Host frames (nrf_group_submit_host_u8: what ngp::NerfRender::render_frame returns): the members render their shards
// as PACKED 8-bit pixels (4 B/px on the links instead of 20), the first member untiles them straight into the planar
// layout of the reference's Image, and one asynchronous copy per plane brings the batch into pinned host memory; two
// slots, so that this copy overlaps the next call's render.
// Built only on the public ABI plus HIP peer copies / RCCL point-to-point calls.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is opened with dlopen (rccl_api)

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/nerfhip.h"

extern "C" void nrf_set_last_error_(const char* msg);

namespace {
int gfail(int code, const std::string& msg) {
  nrf_set_last_error_(msg.c_str());
  return code;
}
#define GHIP(expr)                                                                        \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) return gfail(NRF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define GTRY(expr)          \
  do {                      \
    int _rc = (expr);       \
    if (_rc != NRF_OK) return _rc; \
  } while (0)

// librccl's entry points (the point-to-point subset a direct gather needs), resolved once per process
struct RcclApi {
  void* so = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};
const RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.so) break;
    }
    if (!api.so) {
      const char* e = dlerror();
      api.error = std::string("librccl could not be opened: ") + (e ? e : "?");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* n) {
      void* p = dlsym(api.so, n);
      if (!p) {
        ok = false;
        api.error = std::string("librccl lacks ") + n;
      }
      return p;
    };
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
      dlclose(api.so);
      api.so = nullptr;
    }
  });
  return api;
}
#define GNCCL(expr)                                                                                         \
  do {                                                                                                      \
    ncclResult_t _r = (expr);                                                                               \
    if (_r != ncclSuccess) return gfail(NRF_E_HIP, std::string(#expr) + ": " + rccl_api().GetErrorString(_r)); \
  } while (0)
}  // namespace

struct nrf_group {
  std::vector<int> devices;
  std::vector<nrf_context*> ctx;
  int gather = NRF_GATHER_PEER_COPY;
  std::vector<ncclComm_t> comm;     // NRF_GATHER_RCCL: member i's communicator (rank i of n)
  int rccl_version = 0;
  std::vector<hipStream_t> stream;  // one per member, on its device
  std::vector<hipEvent_t> done;     // shard of member i has arrived on devices[0]
  nrf_options opt{};
  int W = 0, H = 0, tps = 0, max_views = 0, last_views = 0;
  void* gathered_rgba = nullptr;  // on devices[0]: [member][view][tps*64][4] f32
  void* gathered_depth = nullptr; //                [member][view][tps*64]    f32
  void* frame_rgba = nullptr;     // on devices[0]: [view][H][W][4]
  void* frame_depth = nullptr;    //                [view][H][W]
  void* scratch8 = nullptr;       // nrf_group_read_view_u8 scratch on devices[0]: rgb u8 [H][W][3] | depth u8 [H][W]
  // host frames
  struct HostSlot {
    void* gathered = nullptr;      // devices[0]: [member][view][tps*64] packed pixels
    void* d_buf = nullptr;         // devices[0]: rgb u8 [views][px][3] | depth u8 [views][px]
    uint8_t* h_buf = nullptr;      // pinned host, same layout
    hipEvent_t untiled = nullptr, done = nullptr, t0 = nullptr;  // t0 .. untiled: the call's device time on devices[0]
    int n_views = 0, views = 0;    // views of the last call / views the slot's buffers hold
    bool with_depth = true, pending = false, single = false;
    int member_ticket = -1;        // single-member groups: the member context's own ticket
  } hs[2];
  int hs_next = 0, shard_views = 0;  // shard_views: views the members' shard buffers hold
  std::vector<void*> shard8;        // member i, on its device: [view][tps*64] packed pixels
  hipStream_t copy_stream = nullptr;  // on devices[0]
};

namespace {
// does a call go through the exchange step (shards -> devices[0] -> untile)?  A lone member renders row-major frames itself --
// unless the group was asked for the RCCL transport: then even one member runs the whole exchange
bool exchanges(const nrf_group* g) { return g->ctx.size() > 1 || g->gather == NRF_GATHER_RCCL; }

// The exchange step: for every segment s (a plane of the batch) member i's `bytes[s]` at src[s][i] (on its device, written by
// work queued on its stream) go to dst[s] + i * bytes[s] on devices[0]; when this returns, work queued on stream[0]
// afterwards sees all of it.
struct Segment {
  std::vector<const void*> src;  // per member
  char* dst;
  size_t bytes;
};
int ship_to_first(nrf_group* g, const std::vector<Segment>& segs) {
  const size_t n = g->ctx.size();
  if (g->gather == NRF_GATHER_RCCL) {
    const RcclApi& R = rccl_api();
    GNCCL(R.GroupStart());
    for (size_t i = 0; i < n; ++i)
      for (const Segment& sg : segs) {
        ncclResult_t r = R.Send(sg.src[i], sg.bytes, ncclUint8, 0, g->comm[i], g->stream[i]);
        if (r == ncclSuccess) r = R.Recv(sg.dst + i * sg.bytes, sg.bytes, ncclUint8, (int)i, g->comm[0], g->stream[0]);
        if (r != ncclSuccess) {
          (void)R.GroupEnd();
          return gfail(NRF_E_HIP, std::string("ncclSend / ncclRecv: ") + R.GetErrorString(r));
        }
      }
    GNCCL(R.GroupEnd());
    return NRF_OK;
  }
  for (size_t i = 0; i < n; ++i) {
    GHIP(hipSetDevice(g->devices[i]));
    for (const Segment& sg : segs)
      GHIP(hipMemcpyPeerAsync(sg.dst + i * sg.bytes, g->devices[0], sg.src[i], g->devices[i], sg.bytes, g->stream[i]));
    GHIP(hipEventRecord(g->done[i], g->stream[i]));
  }
  GHIP(hipSetDevice(g->devices[0]));
  for (size_t i = 1; i < n; ++i) GHIP(hipStreamWaitEvent(g->stream[0], g->done[i], 0));
  return NRF_OK;
}

void destroy_comms(nrf_group* g) {
  for (size_t i = 0; i < g->comm.size(); ++i)
    if (g->comm[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)rccl_api().CommDestroy(g->comm[i]);
    }
  g->comm.clear();
}

void free_buffers(nrf_group* g) {
  if (g->devices.empty()) return;
  (void)hipSetDevice(g->devices[0]);
  for (void** p : {&g->gathered_rgba, &g->gathered_depth, &g->frame_rgba, &g->frame_depth, &g->scratch8}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
}

void free_host_slots(nrf_group* g) {
  if (g->devices.empty()) return;
  for (size_t i = 0; i < g->shard8.size(); ++i) {
    (void)hipSetDevice(g->devices[i]);
    if (g->shard8[i]) (void)hipFree(g->shard8[i]);
    g->shard8[i] = nullptr;
  }
  (void)hipSetDevice(g->devices[0]);
  for (auto& h : g->hs) {
    if (h.gathered) (void)hipFree(h.gathered);
    if (h.d_buf) (void)hipFree(h.d_buf);
    if (h.h_buf) (void)hipHostFree(h.h_buf);
    h.gathered = h.d_buf = nullptr;
    h.h_buf = nullptr;
    h.pending = false;
    h.views = 0;
  }
  g->shard_views = 0;
}

int ensure_buffers(nrf_group* g, int n_views) {
  if (n_views <= g->max_views && (g->frame_rgba || !exchanges(g))) return NRF_OK;
  const size_t n = g->ctx.size();
  for (nrf_context* c : g->ctx) GTRY(nrf_set_max_views(c, n_views));
  g->max_views = n_views;
  if (!exchanges(g)) return NRF_OK;  // a single member renders row-major frames itself
  free_buffers(g);
  GHIP(hipSetDevice(g->devices[0]));
  const size_t shard_px = (size_t)g->tps * 64, frame_px = (size_t)g->W * g->H;
  GHIP(hipMalloc(&g->gathered_rgba, n * n_views * shard_px * 16));
  GHIP(hipMalloc(&g->gathered_depth, n * n_views * shard_px * 4));
  GHIP(hipMalloc(&g->frame_rgba, (size_t)n_views * frame_px * 16));
  GHIP(hipMalloc(&g->frame_depth, (size_t)n_views * frame_px * 4));
  GHIP(hipMalloc(&g->scratch8, frame_px * 4));
  return NRF_OK;
}
}  // namespace

extern "C" {

int nrf_group_create(int n_devices, const int* devices, nrf_group** out) {
  if (!out || n_devices < 1) return gfail(NRF_E_INVALID, "nrf_group_create: bad argument");
  nrf_group* g = new nrf_group;
  nrf_default_options(&g->opt);
  for (int i = 0; i < n_devices; ++i) g->devices.push_back(devices ? devices[i] : i);
  for (int i = 0; i < n_devices; ++i) {
    nrf_context* c = nullptr;
    const int rc = nrf_create(g->devices[i], &c);
    if (rc != NRF_OK) {
      nrf_group_destroy(g);
      return rc;
    }
    g->ctx.push_back(c);
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      nrf_group_destroy(g);
      return gfail(NRF_E_HIP, "nrf_group_create: stream / event creation failed");
    }
    g->stream.push_back(s);
    g->done.push_back(e);
    // peer access from the first device to this one (and back): without it the copies are staged through the host
    if (g->devices[i] != g->devices[0]) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->devices[i], g->devices[0]) == hipSuccess && can) {
        (void)hipDeviceEnablePeerAccess(g->devices[0], 0);  // current device is devices[i]
        (void)hipGetLastError();                            // "already enabled" is fine
      }
    }
  }
  g->shard8.assign(g->ctx.size(), nullptr);
  // (a stream of another priority: a hardware queue it shares with no render stream -- see nrf_create)
  int prio_lo = 0, prio_hi = 0;
  bool ok = hipSetDevice(g->devices[0]) == hipSuccess && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess &&
            hipStreamCreateWithPriority(&g->copy_stream, hipStreamNonBlocking, prio_hi) == hipSuccess;
  for (auto& h : g->hs)
    ok = ok && hipEventCreate(&h.untiled) == hipSuccess && hipEventCreate(&h.t0) == hipSuccess &&
         hipEventCreateWithFlags(&h.done, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    nrf_group_destroy(g);
    return gfail(NRF_E_HIP, "nrf_group_create: stream / event creation failed");
  }
  // NRF_GROUP_GATHER=rccl: the RCCL transport for callers that only know the reference's API (NerfRender(n), render_server's
  // tile mode); anything else than "rccl" / "peer" is an error rather than a silent default
  if (const char* e = std::getenv("NRF_GROUP_GATHER")) {
    const std::string v(e);
    if (v != "rccl" && v != "peer" && !v.empty()) {
      nrf_group_destroy(g);
      return gfail(NRF_E_INVALID, "NRF_GROUP_GATHER must be rccl or peer");
    }
    if (v == "rccl") {
      const int rc = nrf_group_set_gather(g, NRF_GATHER_RCCL);
      if (rc != NRF_OK) {
        nrf_group_destroy(g);
        return rc;
      }
    }
  }
  *out = g;
  return NRF_OK;
}

int nrf_group_set_gather(nrf_group* g, int mode) {
  if (!g || (mode != NRF_GATHER_PEER_COPY && mode != NRF_GATHER_RCCL)) return gfail(NRF_E_INVALID, "nrf_group_set_gather: bad argument");
  if (mode == g->gather) return NRF_OK;
  // the switch frees the host-frame slots (set_resolution below): a ticket that has not been waited for would be dropped, and its
  // copy would land in freed pinned memory
  for (const auto& h : g->hs)
    if (h.pending) return gfail(NRF_E_STATE, "nrf_group_set_gather: a host-frame ticket is outstanding (nrf_group_wait_host_u8 first)");
  const size_t n = g->ctx.size();
  for (size_t i = 0; i < n; ++i) {  // nothing of the other transport is left in flight
    GHIP(hipSetDevice(g->devices[i]));
    GHIP(hipStreamSynchronize(g->stream[i]));
  }
  if (mode == NRF_GATHER_RCCL) {
    const RcclApi& R = rccl_api();
    if (!R.so) return gfail(NRF_E_UNSUPPORTED, "nrf_group_set_gather: " + R.error);
    if (std::set<int>(g->devices.begin(), g->devices.end()).size() != n)
      return gfail(NRF_E_UNSUPPORTED, "nrf_group_set_gather: RCCL needs one rank per DISTINCT device (a group that lists a device "
                                      "twice is a rehearsal: it keeps the peer copies)");
    std::vector<ncclComm_t> comm(n, nullptr);
    GNCCL(R.CommInitAll(comm.data(), (int)n, g->devices.data()));
    g->comm = comm;
    (void)R.GetVersion(&g->rccl_version);
  } else {
    destroy_comms(g);
  }
  g->gather = mode;
  // a lone member changes layout with the transport (row-major <-> its shard): options and buffers follow
  GTRY(nrf_group_set_options(g, &g->opt));
  if (g->W > 0) GTRY(nrf_group_set_resolution(g, g->W, g->H));
  return NRF_OK;
}

int nrf_group_get_gather(const nrf_group* g, int* mode, int* rccl_version) {
  if (!g) return gfail(NRF_E_INVALID, "null argument");
  if (mode) *mode = g->gather;
  if (rccl_version) *rccl_version = g->gather == NRF_GATHER_RCCL ? g->rccl_version : 0;
  return NRF_OK;
}

int nrf_group_destroy(nrf_group* g) {
  if (!g) return NRF_OK;
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    (void)hipSetDevice(g->devices[i]);
    (void)hipDeviceSynchronize();
  }
  free_buffers(g);
  free_host_slots(g);
  destroy_comms(g);
  if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
  for (auto& h : g->hs) {
    if (h.untiled) (void)hipEventDestroy(h.untiled);
    if (h.t0) (void)hipEventDestroy(h.t0);
    if (h.done) (void)hipEventDestroy(h.done);
  }
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  for (size_t i = 0; i < g->stream.size(); ++i) {
    (void)hipSetDevice(g->devices[i]);
    if (g->done[i]) (void)hipEventDestroy(g->done[i]);
    if (g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
  }
  for (nrf_context* c : g->ctx) (void)nrf_destroy(c);
  delete g;
  return NRF_OK;
}

int nrf_group_size(const nrf_group* g) { return g ? (int)g->ctx.size() : 0; }

nrf_context* nrf_group_member(nrf_group* g, int index) {
  return (g && index >= 0 && index < (int)g->ctx.size()) ? g->ctx[index] : nullptr;
}

int nrf_group_load_model(nrf_group* g, const nrf_model_desc* d) {
  if (!g || !d) return gfail(NRF_E_INVALID, "null argument");
  for (nrf_context* c : g->ctx) GTRY(nrf_load_model(c, d));  // replicated: 24 MB table + occupancy per device
  return NRF_OK;
}

int nrf_group_set_options(nrf_group* g, const nrf_options* o) {
  if (!g || !o) return gfail(NRF_E_INVALID, "null argument");
  g->opt = *o;
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    nrf_options m = *o;
    m.shard_index = (int)i;  // the partition is the group's business
    m.shard_count = (int)g->ctx.size();
    m.tile_major = exchanges(g) ? 1 : 0;  // (what a lone member of an RCCL group needs said; implied for two and more)
    GTRY(nrf_set_options(g->ctx[i], &m));
  }
  return NRF_OK;
}

int nrf_group_set_resolution(nrf_group* g, int width, int height) {
  if (!g || width <= 0 || height <= 0) return gfail(NRF_E_INVALID, "bad resolution");
  GTRY(nrf_group_set_options(g, &g->opt));
  for (nrf_context* c : g->ctx) GTRY(nrf_set_resolution(c, width, height));
  g->W = width;
  g->H = height;
  GTRY(nrf_tiles_per_shard(width, height, (int)g->ctx.size(), &g->tps));
  free_buffers(g);
  free_host_slots(g);
  g->max_views = 0;
  return ensure_buffers(g, 1);
}

int nrf_group_render_views(nrf_group* g, int n_views, const float* cams, const float* poses, nrf_frame* out) {
  if (!g || !cams || !poses || n_views < 1) return gfail(NRF_E_INVALID, "bad argument");
  if (g->W <= 0) return gfail(NRF_E_STATE, "nrf_group_set_resolution has not been called");
  GTRY(ensure_buffers(g, n_views));
  const size_t n = g->ctx.size();
  g->last_views = n_views;
  if (!exchanges(g)) {
    GTRY(nrf_render_views(g->ctx[0], n_views, cams, poses, (void*)g->stream[0], out));
    GHIP(hipStreamSynchronize(g->stream[0]));
    return NRF_OK;
  }
  const size_t shard_px = (size_t)g->tps * 64;
  std::vector<Segment> segs(2);
  segs[0] = {std::vector<const void*>(n), (char*)g->gathered_rgba, (size_t)n_views * shard_px * 16};
  segs[1] = {std::vector<const void*>(n), (char*)g->gathered_depth, (size_t)n_views * shard_px * 4};
  for (size_t i = 0; i < n; ++i) {  // every device renders its shards on its own stream
    nrf_frame f;
    GTRY(nrf_render_views(g->ctx[i], n_views, cams, poses, (void*)g->stream[i], &f));
    segs[0].src[i] = f.rgba;
    segs[1].src[i] = f.depth;
  }
  GTRY(ship_to_first(g, segs));
  GHIP(hipSetDevice(g->devices[0]));
  GTRY(nrf_untile_views(g->ctx[0], g->gathered_rgba, (int)n, g->tps, 4, n_views, g->frame_rgba, (void*)g->stream[0]));
  GTRY(nrf_untile_views(g->ctx[0], g->gathered_depth, (int)n, g->tps, 1, n_views, g->frame_depth, (void*)g->stream[0]));
  GHIP(hipStreamSynchronize(g->stream[0]));
  if (out) {
    out->width = g->W;
    out->height = g->H;
    out->n_tiles = ((g->W + 7) / 8) * ((g->H + 7) / 8);
    out->rgba = g->frame_rgba;
    out->depth = g->frame_depth;
    out->tile_major = 0;
    out->n_views = n_views;
    out->view_stride_px = (int64_t)g->W * g->H;
  }
  return NRF_OK;
}

int nrf_group_read_view_f32(nrf_group* g, int view, float* rgba, float* depth) {
  if (!g || view < 0 || view >= g->last_views) return gfail(NRF_E_INVALID, "view index out of range");
  if (!exchanges(g)) return nrf_read_view_f32(g->ctx[0], view, rgba, depth);
  const size_t px = (size_t)g->W * g->H;
  GHIP(hipSetDevice(g->devices[0]));
  if (rgba) GHIP(hipMemcpy(rgba, (const char*)g->frame_rgba + (size_t)view * px * 16, px * 16, hipMemcpyDeviceToHost));
  if (depth) GHIP(hipMemcpy(depth, (const char*)g->frame_depth + (size_t)view * px * 4, px * 4, hipMemcpyDeviceToHost));
  return NRF_OK;
}

int nrf_group_read_view_u8(nrf_group* g, int view, uint8_t* rgb, uint8_t* depth) {
  if (!g || view < 0 || view >= g->last_views) return gfail(NRF_E_INVALID, "view index out of range");
  if (!exchanges(g)) return nrf_read_view_u8(g->ctx[0], view, rgb, depth);
  const size_t px = (size_t)g->W * g->H;
  GHIP(hipSetDevice(g->devices[0]));
  // quantised into the Image layout on the device (nerf_render.cu:352-359 is a host loop); the copies go to the caller's memory
  uint8_t* d_rgb = (uint8_t*)g->scratch8;
  uint8_t* d_depth = d_rgb + px * 3;
  GTRY(nrf_quantize_u8(g->ctx[0], (const char*)g->frame_rgba + (size_t)view * px * 16, (const char*)g->frame_depth + (size_t)view * px * 4,
                       px, d_rgb, d_depth, (void*)g->stream[0]));
  GHIP(hipStreamSynchronize(g->stream[0]));
  if (rgb) GHIP(hipMemcpy(rgb, d_rgb, px * 3, hipMemcpyDeviceToHost));
  if (depth) GHIP(hipMemcpy(depth, d_depth, px, hipMemcpyDeviceToHost));
  return NRF_OK;
}

// ---- host frames (nerfhip.h "host frames")
}  // extern "C"
namespace {
// issues every launch and copy of a group submit into slot h; the caller commits the slot (ticket, n_views, hs_next) only
// when this has returned NRF_OK
int group_submit_impl(nrf_group* g, nrf_group::HostSlot& h, int n_views, const float* cams, const float* poses, int flags) {
  const size_t n = g->ctx.size();
  h.with_depth = !(flags & NRF_HOST_RGB_ONLY);
  if (!exchanges(g)) {  // one member: its own host-frame path (region-of-interest rows only, no untile)
    h.single = true;
    const int rc = nrf_submit_host_u8(g->ctx[0], n_views, cams, poses, flags, &h.member_ticket);
    h.pending = rc == NRF_OK;
    return rc;
  }
  h.single = false;
  GHIP(hipSetDevice(g->devices[0]));
  if (h.pending) GHIP(hipEventSynchronize(h.done));
  h.pending = false;
  const size_t shard_px = (size_t)g->tps * 64, frame_px = (size_t)g->W * g->H;
  if (n_views > g->shard_views) {  // the members' shard buffers (their streams are drained first: a render may still write them)
    for (size_t i = 0; i < n; ++i) {
      GHIP(hipSetDevice(g->devices[i]));
      GHIP(hipStreamSynchronize(g->stream[i]));
      if (g->shard8[i]) (void)hipFree(g->shard8[i]);
      g->shard8[i] = nullptr;
      GHIP(hipMalloc(&g->shard8[i], (size_t)n_views * shard_px * 4));
    }
    g->shard_views = n_views;
    GHIP(hipSetDevice(g->devices[0]));
  }
  if (n_views > h.views) {  // this slot only: the other one may hold frames that have not been read yet
    GHIP(hipStreamSynchronize(g->stream[0]));
    GHIP(hipStreamSynchronize(g->copy_stream));
    if (h.gathered) (void)hipFree(h.gathered);
    if (h.d_buf) (void)hipFree(h.d_buf);
    if (h.h_buf) (void)hipHostFree(h.h_buf);
    h.gathered = h.d_buf = nullptr;
    h.h_buf = nullptr;
    void* hp = nullptr;
    GHIP(hipMalloc(&h.gathered, n * n_views * shard_px * 4));
    GHIP(hipMalloc(&h.d_buf, (size_t)n_views * frame_px * 4));
    GHIP(hipHostMalloc(&hp, (size_t)n_views * frame_px * 4, hipHostMallocPortable));
    h.h_buf = (uint8_t*)hp;
    h.views = n_views;
  }
  GHIP(hipSetDevice(g->devices[0]));
  GHIP(hipEventRecord(h.t0, g->stream[0]));
  std::vector<Segment> segs(1);
  segs[0] = {std::vector<const void*>(n), (char*)h.gathered, (size_t)n_views * shard_px * 4};
  for (size_t i = 0; i < n; ++i) {  // every member renders its packed shard on its own stream
    GTRY(nrf_bind_output_rgbd8(g->ctx[i], g->shard8[i]));
    const int rc = nrf_render_views(g->ctx[i], n_views, cams, poses, (void*)g->stream[i], nullptr);
    (void)nrf_bind_output_rgbd8(g->ctx[i], nullptr);
    if (rc != NRF_OK) return rc;
    segs[0].src[i] = g->shard8[i];
  }
  GTRY(ship_to_first(g, segs));
  GHIP(hipSetDevice(g->devices[0]));
  uint8_t* d_rgb = (uint8_t*)h.d_buf;
  uint8_t* d_depth = d_rgb + (size_t)h.views * frame_px * 3;
  GTRY(nrf_untile_views_u8(g->ctx[0], h.gathered, (int)n, g->tps, n_views, d_rgb, d_depth, (void*)g->stream[0]));
  GHIP(hipEventRecord(h.untiled, g->stream[0]));
  GHIP(hipStreamWaitEvent(g->copy_stream, h.untiled, 0));
  GHIP(hipMemcpyAsync(h.h_buf, d_rgb, (size_t)n_views * frame_px * 3, hipMemcpyDeviceToHost, g->copy_stream));
  if (h.with_depth)
    GHIP(hipMemcpyAsync(h.h_buf + (size_t)h.views * frame_px * 3, d_depth, (size_t)n_views * frame_px, hipMemcpyDeviceToHost,
                        g->copy_stream));
  GHIP(hipEventRecord(h.done, g->copy_stream));
  h.pending = true;
  g->last_views = 0;  // (the float frames of nrf_group_render_views are not what was rendered last)
  return NRF_OK;
}
}  // namespace
extern "C" {

int nrf_group_submit_host_u8(nrf_group* g, int n_views, const float* cams, const float* poses, int flags, int* ticket) {
  if (!g || !cams || !poses || !ticket || n_views < 1) return gfail(NRF_E_INVALID, "bad argument");
  if (g->W <= 0) return gfail(NRF_E_STATE, "nrf_group_set_resolution has not been called");
  const int si = g->hs_next;
  nrf_group::HostSlot& h = g->hs[si];
  const int rc = group_submit_impl(g, h, n_views, cams, poses, flags);
  if (rc != NRF_OK) {
    // nothing is committed: the slot holds no frames (a wait on it reports NRF_E_STATE instead of stale bytes), the next
    // submit takes the same slot again, and whatever a member has already been handed (a render, half of the peer copies)
    // is drained so that no copy is left in flight into buffers the next submit may reallocate
    h.n_views = 0;
    h.pending = false;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
      if (hipSetDevice(g->devices[i]) == hipSuccess && i < g->stream.size() && g->stream[i]) (void)hipStreamSynchronize(g->stream[i]);
    }
    if (exchanges(g) && hipSetDevice(g->devices[0]) == hipSuccess && g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
    return rc;
  }
  h.n_views = n_views;
  g->hs_next = (si + 1) % 2;
  *ticket = si;
  return NRF_OK;
}

int nrf_group_wait_host_u8(nrf_group* g, int ticket, nrf_host_frame* out) {
  if (!g || ticket < 0 || ticket >= 2) return gfail(NRF_E_INVALID, "bad ticket");
  nrf_group::HostSlot& h = g->hs[ticket];
  if (h.n_views < 1) return gfail(NRF_E_STATE, "nothing was submitted with this ticket");
  if (h.single) {
    h.pending = false;
    return nrf_wait_host_u8(g->ctx[0], h.member_ticket, out);
  }
  GHIP(hipSetDevice(g->devices[0]));
  if (h.pending) GHIP(hipEventSynchronize(h.done));
  h.pending = false;
  if (out) {
    out->width = g->W;
    out->height = g->H;
    out->n_views = h.n_views;
    out->rgb = h.h_buf;
    out->depth = h.with_depth ? h.h_buf + (size_t)h.views * g->W * g->H * 3 : nullptr;
    out->view_stride_px = (int64_t)g->W * g->H;
    out->render_ms = 0.f;
    out->copied_bytes = (uint64_t)h.n_views * g->W * g->H * (h.with_depth ? 4 : 3);
    (void)hipEventElapsedTime(&out->render_ms, h.t0, h.untiled);
  }
  return NRF_OK;
}

int nrf_group_render_host_u8(nrf_group* g, int n_views, const float* cams, const float* poses, int flags, nrf_host_frame* out) {
  int ticket = -1;
  GTRY(nrf_group_submit_host_u8(g, n_views, cams, poses, flags, &ticket));
  return nrf_group_wait_host_u8(g, ticket, out);
}

int nrf_group_get_stats(nrf_group* g, nrf_stats* s) {
  if (!g || !s) return gfail(NRF_E_INVALID, "null argument");
  std::memset(s, 0, sizeof(*s));
  for (nrf_context* c : g->ctx) {
    nrf_stats m;
    GTRY(nrf_get_stats(c, &m));
    s->n_rays += m.n_rays;
    s->n_samples += m.n_samples;
    s->n_rounds += m.n_rounds;
    s->n_network_evals += m.n_network_evals;
    s->n_composited += m.n_composited;
    if (m.render_ms > s->render_ms) s->render_ms = m.render_ms;  // members run concurrently
    s->gather_addresses_per_sample = m.gather_addresses_per_sample;  // (one model, replicated)
    s->grid_device_bytes = m.grid_device_bytes;
    if (m.shader_clock_mhz > 0.f && (s->shader_clock_mhz == 0.f || m.shader_clock_mhz < s->shader_clock_mhz))
      s->shader_clock_mhz = m.shader_clock_mhz;  // (the slowest member's)
  }
  return NRF_OK;
}

}  // extern "C"
