// render_server.cpp -- mirror of the reference's TCP render server (src/render_server.cu:40-108)
// on POSIX sockets (sockpp is not vendored).  Wire protocol, unchanged: the client sends 16
// little-endian f32 = row-major 4x4 camera-to-world; the server answers exactly 3*W*H bytes of
// RGB u8, row-major, no header.  Fixed camera {840, 840, 339, 590} and 1080x1080 as in the reference.
//
// Unlike the reference (one client at a time, one render_frame per request) any number of clients
// may be connected.  Each connection has a reader thread that hands its pose to a DISPATCHER, which
// deals whole requests to per-GPU queues (the queue with the fewest views waiting or rendering:
// replica-parallel, every GPU holds the model -- BASELINE config 5, SURVEY 8(f)1 "per-GPU queues").
// Every GPU has ONE worker thread that owns a single-device NerfRender and runs a two-stage
// pipeline on the two host-frame slots of the C ABI:
//     take up to SERVER_BATCH_VIEWS (64) queued poses -> submit_frames (one launch of the fused kernel + an
//     asynchronous copy of the 8-bit images into pinned host memory) -> ONLY THEN wait for the PREVIOUS batch's
//     copy and wake its clients,
// so batch k + 1 renders while batch k is copied, handed over and sent: the GPU is never idle while a
// reply is in flight.  Requests that arrive while a batch renders form the next batch, so batching needs no
// timer and a lone client sees no added latency.  A reply is sent straight from the pinned slot (no
// per-request copy).  A slow consumer never holds a slot (and with it the GPU): the image being sent has its remainder
// copied out as soon as the socket stops taking bytes, and the images of the same connection that wait behind it are
// copied out the moment the worker wants their slot back (`SlotUse::wanted`, looked at every 2 ms by every connection
// thread that waits for a socket or for a render) -- within a global budget of spilled bytes (NRF_SERVER_SPILL_MB, default
// 1024); a connection that would exceed it, or that takes no byte for NRF_SERVER_SEND_TIMEOUT_S (default 30), is dropped.
// NERF_SERVER_MODE=tile keeps the reference's own multi-GPU form instead: one NerfRender over all devices,
// every frame tile-sharded over them (NGPU of common.h:91).
// Devices: NERF_DEVICES="0,1,..." (repeats allowed: "0,0" rehearses two workers on one GPU), else
// NERF_NGPU=n -> 0..n-1, else device 0.
// Extended request (optional, same connection): the 4 bytes "NRF1", u32 n, then n x {f32 cam[4] = fl_x, fl_y,
// cx, cy; f32 pose[16]}; the answer is n images of 3*W*H bytes in request order.  A raw 64-byte pose keeps
// meaning what it means to the reference's clients.
// Robustness (the reference relies on sockpp::socket_initializer for the first point and has none of the others):
//   * a client that disconnects mid-reply must not take the server down: SIGPIPE is ignored and every send uses
//     MSG_NOSIGNAL;
//   * an extended request of n views is served in chunks of SERVER_BATCH_VIEWS: a connection never has more than that
//     many views queued or in flight, whatever n (<= NRF1_MAX_VIEWS_PER_REQUEST) says;
//   * at most NRF_SERVER_MAX_CLIENTS (default 256) connections are served at a time, further ones are closed at once;
//     client threads are detached and share ownership of the server state, so none can outlive it;
//   * NRF_SERVER_BIND=<ipv4> restricts the listening address (default: any, like the reference's acceptor);
//   * the "QUIT" message (prints the batch statistics and stops the server) and the "STAT" message (answers with one
//     text line of the same statistics, 256 bytes, zero-padded) are test hooks: honoured only when
//     NRF_SERVER_TEST_HOOKS=1 is set in the server's environment.
//   usage: render_server [port=12345] [snapshot=./freality.msgpack] [width height]
#include <arpa/inet.h>
#include <csignal>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nerf_render.h"

using namespace ngp;

static bool read_n(int fd, void* buf, size_t n) {  // the reference does not handle partial reads; this does
  char* p = (char*)buf;
  while (n) {
    const ssize_t r = ::read(fd, p, n);
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}
static bool write_n(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    const ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);  // a vanished peer is an error code, not a SIGPIPE
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}

namespace {

constexpr uint32_t NRF1_MAX_VIEWS_PER_REQUEST = 4096;
constexpr uint32_t SERVER_BATCH_VIEWS = 64;  // frames one launch of a worker renders (BASELINE config 5's 64 requests)
static_assert(SERVER_BATCH_VIEWS <= NRF_MAX_VIEWS, "a batch is one launch");

// A worker's host-frame slot (two per worker, as the C ABI has): how many of its images are still being sent.
struct SlotUse {
  std::mutex m;
  std::condition_variable cv;
  int readers = 0;
  std::atomic<bool> wanted{false};  // the worker waits for this slot: connection threads copy their pending images out of it
};

struct Request {
  Camera cam;
  Matrix4f pose;
  const unsigned char* rgb = nullptr;  // set by the worker: this request's image inside its pinned host-frame slot
  SlotUse* slot = nullptr;             // released by the client thread once the image has left the slot
  bool done = false, failed = false;
  std::mutex m;
  std::condition_variable cv;
};

struct Worker {
  int device = 0;
  std::unique_ptr<NerfRender> render;
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Request>> queue;
  std::atomic<int> load{0};  // views queued or being rendered
  SlotUse slot[2];
  std::thread thread;
  // statistics
  std::atomic<unsigned long> batches{0}, frames{0};
  std::atomic<unsigned long long> gpu_us{0};  // device time of the launches (event-timed, nrf_stats::render_ms)
};

struct Server {
  std::vector<std::unique_ptr<Worker>> workers;
  std::atomic<bool> stop{false};
  std::atomic<int> live_clients{0};
  int max_clients = 256;
  bool test_hooks = false;  // NRF_SERVER_TEST_HOOKS=1
  size_t frame_bytes = 0;
  // images copied out of a slot on behalf of slow consumers: bounded, a connection that would exceed the budget is dropped
  std::atomic<long long> spill_bytes{0};
  long long spill_budget = 1024ll << 20;  // NRF_SERVER_SPILL_MB
  int send_timeout_ms = 30000;            // NRF_SERVER_SEND_TIMEOUT_S: a client that takes no byte for this long is dropped
  std::atomic<unsigned long> evictions{0}, dropped_slow{0};
  int listen_fd = -1;
  std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();

  std::string stat_line() const {
    unsigned long b = 0, f = 0;
    unsigned long long us = 0;
    for (const auto& w : workers) {
      b += w->batches.load();
      f += w->frames.load();
      us += w->gpu_us.load();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    char buf[256];
    std::snprintf(buf, sizeof(buf), "batches %lu frames %lu workers %zu gpu_ms %.3f wall_ms %.3f evictions %lu dropped_slow %lu", b, f,
                  workers.size(), (double)us * 1e-3, wall_ms, evictions.load(), dropped_slow.load());
    return buf;
  }
};

void finish_batch(Worker& w, std::vector<std::shared_ptr<Request>>& batch, int ticket, int slot_index, bool ok) {
  std::vector<Image> imgs;
  if (ok) {
    try {
      imgs = w.render->wait_frames(ticket);
      w.gpu_us += (unsigned long long)(w.render->last_wait_render_ms() * 1e3);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "render error: %s\n", e.what());
      ok = false;
    }
  }
  SlotUse& su = w.slot[slot_index];
  if (ok) {
    std::lock_guard<std::mutex> lk(su.m);
    su.readers += (int)batch.size();
  }
  w.batches++;
  w.frames += batch.size();
  w.load -= (int)batch.size();
  for (size_t i = 0; i < batch.size(); ++i) {
    Request& r = *batch[i];
    std::lock_guard<std::mutex> lk(r.m);
    r.rgb = ok ? imgs[i].rgb : nullptr;
    r.slot = ok ? &su : nullptr;
    r.failed = !ok;
    r.done = true;
    r.cv.notify_one();
  }
  batch.clear();
}

// the one thread that owns a GPU's renderer: submit batch k + 1, then complete batch k
void worker_loop(Server& s, Worker& w) {
  std::vector<std::shared_ptr<Request>> prev, batch;
  int prev_ticket = -1, prev_slot = 0, next_slot = 0;
  while (true) {
    batch.clear();
    {
      std::unique_lock<std::mutex> lk(w.m);
      // with a batch in flight do not wait for more work: finish that one first
      if (prev.empty()) w.cv.wait(lk, [&] { return s.stop.load() || !w.queue.empty(); });
      while (!w.queue.empty() && batch.size() < (size_t)SERVER_BATCH_VIEWS) {
        batch.push_back(w.queue.front());
        w.queue.pop_front();
      }
    }
    if (batch.empty() && prev.empty()) {
      if (s.stop.load()) return;
      continue;
    }
    int ticket = -1;
    bool ok = true;
    if (!batch.empty()) {
      // the slot this submit overwrites (the ticket it will return): every image of its last batch must have left it;
      // after a failed submit the next ticket is not known: both slots
      for (int si = 0; si < 2; ++si) {
        if (next_slot >= 0 && si != next_slot) continue;
        SlotUse& su = w.slot[si];
        std::unique_lock<std::mutex> lk(su.m);
        if (su.readers != 0) {
          su.wanted = true;  // connection threads holding images of this slot copy them out (<= 2 ms + a memcpy)
          su.cv.wait(lk, [&] { return su.readers == 0; });
          su.wanted = false;
        }
      }
      try {
        std::vector<Camera> cams;
        std::vector<Matrix4f> poses;
        for (const auto& r : batch) {
          cams.push_back(r->cam);
          poses.push_back(r->pose);
        }
        ticket = w.render->submit_frames(cams, poses, /*rgb_only=*/true);
      } catch (const std::exception& e) {
        std::fprintf(stderr, "render error: %s\n", e.what());
        ok = false;
      }
    }
    if (!prev.empty()) finish_batch(w, prev, prev_ticket, prev_slot, true);  // batch k: its copy ran under batch k + 1's render
    if (!batch.empty()) {
      if (!ok) {
        finish_batch(w, batch, -1, 0, false);
        next_slot = -1;
      } else {
        prev.swap(batch);
        prev_ticket = ticket;
        prev_slot = ticket;  // tickets ARE the slot indices of the C ABI's host frames (0, 1, 0, ...)
        next_slot = ticket ^ 1;
      }
    }
  }
}

// stop + wake-up under each worker's mutex: a worker that has just found its predicate false cannot miss the notification
void stop_workers(Server& s) {
  for (const auto& w : s.workers) {
    std::lock_guard<std::mutex> lk(w->m);
    s.stop = true;
    w->cv.notify_all();
  }
}

// the dispatcher: a whole request goes to the GPU with the fewest views waiting or rendering
std::shared_ptr<Request> submit(Server& s, const Camera& cam, const float pose[16]) {
  auto req = std::make_shared<Request>();
  req->cam = cam;
  for (int i = 0; i < 16; ++i) req->pose.m[i] = pose[i];
  Worker* best = s.workers[0].get();
  for (const auto& w : s.workers)
    if (w->load.load() < best->load.load()) best = w.get();
  {
    std::lock_guard<std::mutex> lk(best->m);
    if (s.stop.load()) {  // the workers are leaving (or gone): nobody would ever complete this request
      req->done = req->failed = true;
      return req;
    }
    best->load++;
    best->queue.push_back(req);
  }
  best->cv.notify_one();
  return req;
}

// The replies of one connection, in request order, each sent straight from its pinned slot.  See the file header for what
// happens to a consumer that does not keep up.  Returns false when the connection is lost (or dropped); every slot
// reference of `reqs` has been given back when it returns, whatever happened.
class ReplySender {
 public:
  ReplySender(Server& s, int sock, std::vector<std::shared_ptr<Request>>& reqs) : s_(s), sock_(sock) {
    for (auto& r : reqs) items_.push_back(Item{r, {}, false, false});
  }
  ~ReplySender() {
    for (size_t j = 0; j < items_.size(); ++j) {  // a lost connection: still wait for every image and give its slot back
      wait_done(j, /*service=*/false);
      release(j);
      drop_copy(j);
    }
  }
  bool run() {
    bool ok = true;
    for (cur_ = 0; ok && cur_ < items_.size(); ++cur_) {
      off_ = 0;
      if (!wait_done(cur_, /*service=*/true) || items_[cur_].req->failed) return false;
      ok = send_current();
      release(cur_);
      drop_copy(cur_);
    }
    return ok;
  }

 private:
  struct Item {
    std::shared_ptr<Request> req;
    std::vector<unsigned char> copy;  // the image (the current one: its unsent remainder) once it has left the slot
    bool released, spilled;
  };
  Server& s_;
  int sock_;
  std::vector<Item> items_;
  size_t cur_ = 0, off_ = 0;  // the image being sent and how much of it has gone

  bool is_done(size_t j) {
    std::lock_guard<std::mutex> lk(items_[j].req->m);
    return items_[j].req->done;
  }
  // waits for image j; meanwhile (service) the later images of this connection leave a slot its worker wants back
  bool wait_done(size_t j, bool service) {
    Request& r = *items_[j].req;
    while (true) {
      {
        std::unique_lock<std::mutex> lk(r.m);
        if (r.cv.wait_for(lk, std::chrono::milliseconds(2), [&] { return r.done; })) return true;
      }
      if (service && !service_evictions()) return false;
    }
  }
  void release(size_t j) {
    Item& it = items_[j];
    if (it.released || !it.req->slot) return;
    it.released = true;
    std::lock_guard<std::mutex> lk(it.req->slot->m);
    if (--it.req->slot->readers == 0) it.req->slot->cv.notify_all();
  }
  void drop_copy(size_t j) {
    Item& it = items_[j];
    if (it.spilled) s_.spill_bytes -= (long long)it.copy.size();
    it.spilled = false;
    std::vector<unsigned char>().swap(it.copy);
  }
  // copies what is left of image j out of its slot and gives the slot back; false: over the budget -> drop the connection
  bool spill(size_t j) {
    Item& it = items_[j];
    if (it.released || it.req->failed || !it.req->rgb) return true;
    const size_t from = j == cur_ ? off_ : 0, n = s_.frame_bytes - from;
    if (s_.spill_bytes.fetch_add((long long)n) + (long long)n > s_.spill_budget) {
      s_.spill_bytes -= (long long)n;
      s_.dropped_slow++;
      std::fprintf(stderr, "slow consumer dropped: the budget of spilled reply bytes (NRF_SERVER_SPILL_MB) is used up\n");
      return false;
    }
    it.copy.assign(it.req->rgb + from, it.req->rgb + s_.frame_bytes);
    it.spilled = true;
    release(j);
    return true;
  }
  // the finished images of this connection that sit in a slot its worker waits for leave it now
  bool service_evictions() {
    for (size_t j = cur_; j < items_.size(); ++j) {
      Item& it = items_[j];
      if (it.released || !is_done(j) || it.req->failed || !it.req->slot) continue;
      if (!it.req->slot->wanted.load()) continue;
      if (!spill(j)) return false;
      s_.evictions++;
    }
    return true;
  }
  bool send_current() {
    Item& it = items_[cur_];
    auto last_progress = std::chrono::steady_clock::now();
    while (off_ < s_.frame_bytes) {
      // (an image that was evicted while it waited its turn is sent from its copy: the whole image, offset 0)
      const unsigned char* src = it.spilled ? it.copy.data() + (off_ - (s_.frame_bytes - it.copy.size())) : it.req->rgb + off_;
      const ssize_t r = ::send(sock_, src, s_.frame_bytes - off_, MSG_NOSIGNAL | MSG_DONTWAIT);
      if (r > 0) {
        off_ += (size_t)r;
        last_progress = std::chrono::steady_clock::now();
        continue;
      }
      if (!(r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR))) return false;
      // the socket takes no more for now: what is left of this image leaves the slot (one memcpy), the images behind it when
      // their worker asks; then wait for the socket in 2 ms steps
      if (!it.released && !spill(cur_)) return false;
      if (!service_evictions()) return false;
      pollfd pf{sock_, POLLOUT, 0};
      (void)::poll(&pf, 1, 2);
      if (pf.revents & (POLLERR | POLLHUP | POLLNVAL)) return false;
      if (std::chrono::steady_clock::now() - last_progress > std::chrono::milliseconds(s_.send_timeout_ms)) {
        s_.dropped_slow++;
        std::fprintf(stderr, "slow consumer dropped: no byte taken for %d ms (NRF_SERVER_SEND_TIMEOUT_S)\n", s_.send_timeout_ms);
        return false;
      }
    }
    return true;
  }
};

bool send_replies(Server& s, int sock, std::vector<std::shared_ptr<Request>>& reqs) { return ReplySender(s, sock, reqs).run(); }

void serve_client(int sock, const std::string peer, const Camera default_cam, std::shared_ptr<Server> sp) {
  Server& s = *sp;
  std::cout << "Received a connection request from " << peer << std::endl;
  float nerf_pos[16] = {0};
  while (read_n(sock, nerf_pos, sizeof(nerf_pos))) {
    if (s.test_hooks && std::memcmp(nerf_pos, "QUIT", 4) == 0 && nerf_pos[1] == 0.0f && nerf_pos[15] == 0.0f) {
      std::printf("\n%s\n", s.stat_line().c_str());  // one write: other threads print too
      std::fflush(stdout);
      stop_workers(s);
      ::shutdown(s.listen_fd, SHUT_RDWR);  // wakes the acceptor
      break;
    }
    if (s.test_hooks && std::memcmp(nerf_pos, "STAT", 4) == 0 && nerf_pos[1] == 0.0f && nerf_pos[15] == 0.0f) {
      char line[256] = {0};
      std::snprintf(line, sizeof(line), "%s", s.stat_line().c_str());
      if (!write_n(sock, line, sizeof(line))) break;
      continue;
    }
    if (std::memcmp(nerf_pos, "NRF1", 4) == 0) {  // extended request: the 64 bytes read so far are its first 64
      uint32_t n = 0;
      std::memcpy(&n, (const char*)nerf_pos + 4, 4);
      if (n == 0 || n > NRF1_MAX_VIEWS_PER_REQUEST) break;
      std::vector<float> body((size_t)n * 20);  // 80 bytes per view: <= 320 KB
      const size_t have = sizeof(nerf_pos) - 8, need = body.size() * sizeof(float);
      std::memcpy(body.data(), (const char*)nerf_pos + 8, have < need ? have : need);
      if (need > have && !read_n(sock, (char*)body.data() + have, need - have)) break;
      bool ok = true;
      // chunks of SERVER_BATCH_VIEWS: every chunk is queued before its first wait, sent, and released before the next one is
      // queued -- a connection never has more than SERVER_BATCH_VIEWS views queued or in flight
      for (uint32_t first = 0; ok && first < n; first += SERVER_BATCH_VIEWS) {
        std::vector<std::shared_ptr<Request>> reqs;
        for (uint32_t v = first; v < n && v < first + SERVER_BATCH_VIEWS; ++v) {
          const float* r = body.data() + (size_t)v * 20;
          reqs.push_back(submit(s, Camera{r[0], r[1], r[2], r[3]}, r + 4));
        }
        ok = send_replies(s, sock, reqs);
      }
      if (!ok) break;
      continue;
    }
    std::vector<std::shared_ptr<Request>> one{submit(s, default_cam, nerf_pos)};
    if (!send_replies(s, sock, one)) break;
  }
  std::cout << "Connection closed" << std::endl;
  ::close(sock);
  s.live_clients--;
}

std::vector<int> device_list() {
  std::vector<int> devices;
  if (const char* list = std::getenv("NERF_DEVICES")) {
    for (const char* p = list; *p;) {
      devices.push_back(std::atoi(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  if (devices.empty()) {
    const char* e = std::getenv("NERF_NGPU");
    const int n = e ? std::max(1, std::atoi(e)) : 1;
    for (int i = 0; i < n; ++i) devices.push_back(i);
  }
  return devices;
}

}  // namespace

int main(int argc, char** argv) {
  std::cout << "Hello, Metavese!" << std::endl;
  const int port = argc > 1 ? std::atoi(argv[1]) : 12345;
  const std::string config_path = argc > 2 ? argv[2] : "./freality.msgpack";
  const int W = argc > 4 ? std::atoi(argv[3]) : 1080, H = argc > 4 ? std::atoi(argv[4]) : 1080;
  std::signal(SIGPIPE, SIG_IGN);
  // shared with the (detached) client threads: the state lives until the last of them has let go of it
  auto server = std::make_shared<Server>();
  try {
    Server& s = *server;
    const std::vector<int> devices = device_list();
    const char* mode = std::getenv("NERF_SERVER_MODE");
    const bool tile_mode = mode && std::strcmp(mode, "tile") == 0;
    // replica-parallel (default): one single-device renderer + queue per GPU; tile mode: ONE renderer over all devices
    std::vector<std::vector<int>> groups;
    if (tile_mode) groups.push_back(devices);
    else for (int d : devices) groups.push_back({d});
    for (const auto& g : groups) {
      auto w = std::make_unique<Worker>();
      w->device = g[0];
      w->render = std::make_unique<NerfRender>(g);
      w->render->reload_network_from_file(config_path);  // Init Model
      w->render->set_resolution(Vector2i(W, H));
      s.workers.push_back(std::move(w));
    }
    const float sc = (float)W / 1080.0f;
    const Camera cam = {840 * sc, 840 * sc, 339 * sc, 590 * sc};
    s.frame_bytes = (size_t)3 * W * H;

    const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
    s.listen_fd = srv;
    int one = 1;
    ::setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    if (const char* bind_to = std::getenv("NRF_SERVER_BIND")) {
      if (::inet_pton(AF_INET, bind_to, &addr.sin_addr) != 1) {
        std::cerr << "NRF_SERVER_BIND: not an IPv4 address: " << bind_to << std::endl;
        return 1;
      }
    }
    addr.sin_port = htons((uint16_t)port);
    if (srv < 0 || ::bind(srv, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(srv, 256) != 0) {
      std::cerr << "Error creating the acceptor: " << std::strerror(errno) << std::endl;
      return 1;
    }
    {
      const char* hooks = std::getenv("NRF_SERVER_TEST_HOOKS");
      s.test_hooks = hooks && std::strcmp(hooks, "1") == 0;
      if (const char* mc = std::getenv("NRF_SERVER_MAX_CLIENTS")) s.max_clients = std::max(1, std::atoi(mc));
      if (const char* v = std::getenv("NRF_SERVER_SPILL_MB")) s.spill_budget = (long long)std::max(0, std::atoi(v)) << 20;
      if (const char* v = std::getenv("NRF_SERVER_SEND_TIMEOUT_S")) s.send_timeout_ms = std::max(1, std::atoi(v)) * 1000;
    }
    for (auto& w : s.workers) w->thread = std::thread(worker_loop, std::ref(s), std::ref(*w));
    s.t_start = std::chrono::steady_clock::now();
    std::cout << "Awaiting connections on port " << port << " (" << s.workers.size() << (tile_mode ? " tile-sharded renderer" : " GPU queue(s)")
              << ")..." << std::endl;
    while (!s.stop.load()) {
      sockaddr_in peer{};
      socklen_t len = sizeof(peer);
      const int sock = ::accept(srv, (sockaddr*)&peer, &len);
      if (sock < 0) {
        if (s.stop.load()) break;
        std::cerr << "Error accepting incoming connection: " << std::strerror(errno) << std::endl;
        continue;
      }
      if (s.live_clients.load() >= s.max_clients) {  // every connection costs a thread and up to a batch of queued views
        std::cerr << "connection refused: " << s.max_clients << " clients are being served (NRF_SERVER_MAX_CLIENTS)" << std::endl;
        ::close(sock);
        continue;
      }
      ::setsockopt(sock, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      // detached: a finished connection leaves nothing behind; the thread shares ownership of the server state
      s.live_clients++;
      std::thread(serve_client, sock, std::string(inet_ntoa(peer.sin_addr)), cam, server).detach();
    }
    stop_workers(s);
    for (auto& w : s.workers)
      if (w->thread.joinable()) w->thread.join();
    ::close(srv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
