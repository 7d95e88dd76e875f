// render_server.cpp -- mirror of the reference's TCP render server (src/render_server.cu:40-108)
// on POSIX sockets (sockpp is not vendored).  Wire protocol, unchanged: the client sends 16
// little-endian f32 = row-major 4x4 camera-to-world; the server answers exactly 3*W*H bytes of
// RGB u8, row-major, no header.  Fixed camera {840, 840, 339, 590} and 1080x1080 as in the reference.
//
// Unlike the reference (one client at a time, one render_frame per request) any number of clients
// may be connected: each connection has a reader thread that queues its pose, and ONE render
// thread takes everything that is queued -- up to SERVER_BATCH_VIEWS (32) poses -- into a single
// render_frames call (one launch of the fused kernel).  Requests that arrive while a batch renders
// form the next batch, so batching needs no timer and a lone client sees no added latency
// (BASELINE config 5: many concurrent camera requests).
// Extended request (optional, same connection): the 4 bytes "NRF1", u32 n, then n x {f32 cam[4] = fl_x, fl_y,
// cx, cy; f32 pose[16]}; the answer is n images of 3*W*H bytes in request order.  A raw 64-byte pose keeps
// meaning what it means to the reference's clients.
// Robustness (the reference relies on sockpp::socket_initializer for the first point and has none of the others):
//   * a client that disconnects mid-reply must not take the server down: SIGPIPE is ignored and every send uses
//     MSG_NOSIGNAL;
//   * an extended request of n views is served in chunks of SERVER_BATCH_VIEWS (one launch each): at most that many
//     frames are held per connection, whatever n (<= NRF1_MAX_VIEWS_PER_REQUEST) says;
//   * client threads are detached and counted, nothing grows with the number of connections served;
//   * NRF_SERVER_BIND=<ipv4> restricts the listening address (default: any, like the reference's acceptor);
//   * the "QUIT" message (prints the batch statistics and stops the server) is a test hook: it is honoured only
//     when NRF_SERVER_TEST_HOOKS=1 is set in the server's environment.
//   usage: render_server [port=12345] [snapshot=./freality.msgpack] [width height]
#include <arpa/inet.h>
#include <csignal>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
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

struct Request {
  Camera cam;
  Matrix4f pose;
  std::vector<unsigned char> rgb;  // filled by the render thread
  bool done = false, failed = false;
  std::mutex m;
  std::condition_variable cv;
};

struct Batcher {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::shared_ptr<Request>> queue;
  std::atomic<bool> stop{false};
  std::atomic<unsigned long> batches{0}, frames{0};
  std::atomic<int> live_clients{0};
  bool test_hooks = false;  // NRF_SERVER_TEST_HOOKS=1
};
constexpr uint32_t NRF1_MAX_VIEWS_PER_REQUEST = 4096;
constexpr uint32_t SERVER_BATCH_VIEWS = 32;  // frames one launch of the server renders and holds (<= NRF_MAX_VIEWS)
static_assert(SERVER_BATCH_VIEWS <= NRF_MAX_VIEWS, "a batch is one launch");

// the one thread that owns the renderer
void render_loop(NerfRender& render, const size_t frame_bytes, Batcher& b) {
  while (true) {
    std::vector<std::shared_ptr<Request>> batch;
    {
      std::unique_lock<std::mutex> lk(b.m);
      b.cv.wait(lk, [&] { return b.stop.load() || !b.queue.empty(); });
      if (b.stop.load() && b.queue.empty()) return;
      while (!b.queue.empty() && batch.size() < (size_t)SERVER_BATCH_VIEWS) {
        batch.push_back(b.queue.front());
        b.queue.pop_front();
      }
    }
    bool ok = true;
    try {
      std::vector<Camera> cams;
      std::vector<Matrix4f> poses;
      for (const auto& r : batch) {
        cams.push_back(r->cam);
        poses.push_back(r->pose);
      }
      const std::vector<Image> imgs = render.render_frames(cams, poses);
      for (size_t i = 0; i < batch.size(); ++i) batch[i]->rgb.assign(imgs[i].rgb, imgs[i].rgb + frame_bytes);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "render error: %s\n", e.what());
      ok = false;
    }
    b.batches++;
    b.frames += batch.size();
    for (const auto& r : batch) {
      std::lock_guard<std::mutex> lk(r->m);
      r->done = true;
      r->failed = !ok;
      r->cv.notify_one();
    }
  }
}

std::shared_ptr<Request> submit(Batcher& b, const Camera& cam, const float pose[16]) {
  auto req = std::make_shared<Request>();
  req->cam = cam;
  for (int i = 0; i < 16; ++i) req->pose.m[i] = pose[i];
  {
    std::lock_guard<std::mutex> lk(b.m);
    b.queue.push_back(req);
  }
  b.cv.notify_one();
  return req;
}

bool wait_and_send(int sock, const std::shared_ptr<Request>& req, size_t frame_bytes) {
  {
    std::unique_lock<std::mutex> lk(req->m);
    req->cv.wait(lk, [&] { return req->done; });
  }
  return !req->failed && write_n(sock, req->rgb.data(), frame_bytes);
}

void serve_client(int sock, const std::string peer, const Camera default_cam, const size_t frame_bytes, Batcher& b, int srv) {
  std::cout << "Received a connection request from " << peer << std::endl;
  float nerf_pos[16] = {0};
  while (read_n(sock, nerf_pos, sizeof(nerf_pos))) {
    if (b.test_hooks && std::memcmp(nerf_pos, "QUIT", 4) == 0 && nerf_pos[1] == 0.0f && nerf_pos[15] == 0.0f) {
      std::printf("\nbatches %lu frames %lu\n", b.batches.load(), b.frames.load());  // one write: other threads print too
      std::fflush(stdout);
      b.stop = true;
      b.cv.notify_all();
      ::shutdown(srv, SHUT_RDWR);  // wakes the acceptor
      break;
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
      // chunks of SERVER_BATCH_VIEWS: every chunk is queued before its first wait (one launch), sent, and freed before
      // the next one is queued -- a connection never holds more than SERVER_BATCH_VIEWS rendered frames
      for (uint32_t first = 0; ok && first < n; first += SERVER_BATCH_VIEWS) {
        std::vector<std::shared_ptr<Request>> reqs;
        for (uint32_t v = first; v < n && v < first + SERVER_BATCH_VIEWS; ++v) {
          const float* r = body.data() + (size_t)v * 20;
          reqs.push_back(submit(b, Camera{r[0], r[1], r[2], r[3]}, r + 4));
        }
        for (const auto& r : reqs) ok = ok && wait_and_send(sock, r, frame_bytes);
      }
      if (!ok) break;
      continue;
    }
    if (!wait_and_send(sock, submit(b, default_cam, nerf_pos), frame_bytes)) break;
  }
  std::cout << "Connection closed" << std::endl;
  ::close(sock);
  b.live_clients--;
}

}  // namespace

int main(int argc, char** argv) {
  std::cout << "Hello, Metavese!" << std::endl;
  const int port = argc > 1 ? std::atoi(argv[1]) : 12345;
  const std::string config_path = argc > 2 ? argv[2] : "./freality.msgpack";
  const int W = argc > 4 ? std::atoi(argv[3]) : 1080, H = argc > 4 ? std::atoi(argv[4]) : 1080;
  std::signal(SIGPIPE, SIG_IGN);
  try {
    NerfRender render;
    render.reload_network_from_file(config_path);  // Init Model
    const float s = (float)W / 1080.0f;
    const Camera cam = {840 * s, 840 * s, 339 * s, 590 * s};
    render.set_resolution(Vector2i(W, H));
    const size_t frame_bytes = (size_t)3 * W * H;

    const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
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
    if (srv < 0 || ::bind(srv, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(srv, 64) != 0) {
      std::cerr << "Error creating the acceptor: " << std::strerror(errno) << std::endl;
      return 1;
    }
    Batcher batcher;
    {
      const char* hooks = std::getenv("NRF_SERVER_TEST_HOOKS");
      batcher.test_hooks = hooks && std::strcmp(hooks, "1") == 0;
    }
    std::thread renderer(render_loop, std::ref(render), frame_bytes, std::ref(batcher));
    std::cout << "Awaiting connections on port " << port << "..." << std::endl;
    while (!batcher.stop.load()) {
      sockaddr_in peer{};
      socklen_t len = sizeof(peer);
      const int sock = ::accept(srv, (sockaddr*)&peer, &len);
      if (sock < 0) {
        if (batcher.stop.load()) break;
        std::cerr << "Error accepting incoming connection: " << std::strerror(errno) << std::endl;
        continue;
      }
      ::setsockopt(sock, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      // detached: a finished connection leaves nothing behind (readers of still-open connections end with the process)
      batcher.live_clients++;
      std::thread(serve_client, sock, std::string(inet_ntoa(peer.sin_addr)), cam, frame_bytes, std::ref(batcher), srv).detach();
    }
    batcher.stop = true;
    batcher.cv.notify_all();
    renderer.join();
    ::close(srv);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
