// render_server.cpp -- mirror of the reference's TCP render server (src/render_server.cu:40-108)
// on POSIX sockets (sockpp is not vendored).  Wire protocol, unchanged: the client sends 16
// little-endian f32 = row-major 4x4 camera-to-world; the server answers exactly 3*W*H bytes of
// RGB u8, row-major, no header.  Fixed camera {840, 840, 339, 590} and 1080x1080 as in the reference.
//   usage: render_server [port=12345] [snapshot=./freality.msgpack] [width height]
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

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
    const ssize_t r = ::write(fd, p, n);
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}

int main(int argc, char** argv) {
  std::cout << "Hello, Metavese!" << std::endl;
  const int port = argc > 1 ? std::atoi(argv[1]) : 12345;
  const std::string config_path = argc > 2 ? argv[2] : "./freality.msgpack";
  const int W = argc > 4 ? std::atoi(argv[3]) : 1080, H = argc > 4 ? std::atoi(argv[4]) : 1080;
  try {
    NerfRender render;
    render.reload_network_from_file(config_path);  // Init Model
    const float s = (float)W / 1080.0f;
    Camera cam = {840 * s, 840 * s, 339 * s, 590 * s};
    render.set_resolution(Vector2i(W, H));

    const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    ::setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    addr.sin_port = htons((uint16_t)port);
    if (srv < 0 || ::bind(srv, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(srv, 4) != 0) {
      std::cerr << "Error creating the acceptor: " << std::strerror(errno) << std::endl;
      return 1;
    }
    std::cout << "Awaiting connections on port " << port << "..." << std::endl;
    while (true) {
      sockaddr_in peer{};
      socklen_t len = sizeof(peer);
      const int sock = ::accept(srv, (sockaddr*)&peer, &len);
      if (sock < 0) {
        std::cerr << "Error accepting incoming connection: " << std::strerror(errno) << std::endl;
        continue;
      }
      ::setsockopt(sock, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      std::cout << "Received a connection request from " << inet_ntoa(peer.sin_addr) << std::endl;
      float nerf_pos[16] = {0};
      while (read_n(sock, nerf_pos, sizeof(nerf_pos))) {
        if (std::memcmp(nerf_pos, "QUIT", 4) == 0 && nerf_pos[1] == 0.0f && nerf_pos[15] == 0.0f) {  // test hook
          ::close(sock);
          ::close(srv);
          return 0;
        }
        Matrix4f pose;
        for (int i = 0; i < 16; ++i) pose.m[i] = nerf_pos[i];
        Image img = render.render_frame(cam, pose);
        if (!write_n(sock, img.rgb, (size_t)3 * W * H)) break;
      }
      std::cout << "Connection closed" << std::endl;
      ::close(sock);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
