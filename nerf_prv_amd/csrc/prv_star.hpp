// prv_star.hpp -- a tiny TCP star between the ranks of one job (POSIX sockets, host memory only).
//
// Two uses: (1) the bootstrap of the RCCL communicator in prv_comm.cpp (rank 0's ncclUniqueId reaches the
// other ranks through it -- what torch.distributed's TCP store does for the Python path), and (2) the
// `socket` transport shim of prv_comm: host-staged all-gather / broadcast for the cases RCCL does not cover
// (two ranks sharing one GPU in the tests, CPU-only rendezvous tests).  Not a data path for real multi-GPU
// runs: those gather device buffers with RCCL over xGMI.
//
// Topology: rank 0 listens on <addr>:<port> (MASTER_ADDR / MASTER_PORT + 23 by default, PRV_COMM_PORT
// overrides), every other rank connects (retrying until the listener exists) and introduces itself with its
// rank.  all_gather: every rank sends its block to rank 0, rank 0 answers with the assembled buffer.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace prvstar {

inline bool send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n > 0) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return false; // closed, or no progress within the socket's timeout (EAGAIN)
    c += k;
    n -= (size_t)k;
  }
  return true;
}
inline bool recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n > 0) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return false; // closed, or nothing arrived within the socket's timeout (EAGAIN)
    c += k;
    n -= (size_t)k;
  }
  return true;
}

class Star {
public:
  int rank = 0, world = 1;
  std::string error;

  ~Star() { close_all(); }

  // addr/port: where rank 0 listens.  Empty addr -> $MASTER_ADDR or 127.0.0.1; port <= 0 -> $PRV_COMM_PORT, else
  // $MASTER_PORT + 23 (torchrun keeps its own store on MASTER_PORT), else 29534.
  bool open(int rank_, int world_, std::string addr, int port, double timeout_s = 120.0) {
    rank = rank_;
    world = world_;
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank / world");
    if (world == 1) return true;
    if (addr.empty()) {
      const char* e = getenv("MASTER_ADDR");
      addr = e && *e ? e : "127.0.0.1";
    }
    if (port <= 0) {
      if (const char* e = getenv("PRV_COMM_PORT")) port = atoi(e);
      else if (const char* m = getenv("MASTER_PORT")) port = atoi(m) + 23;
      else port = 29534;
    }
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(addr.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) return fail("cannot resolve " + addr);
    bool ok = rank == 0 ? serve(res, timeout_s) : join(res, timeout_s);
    freeaddrinfo(res);
    return ok;
  }

  // recv = world blocks of `bytes`, rank order, identical on every rank
  bool all_gather(const void* send, size_t bytes, void* recv) {
    if (world == 1) {
      if (recv != send) memcpy(recv, send, bytes);
      return true;
    }
    if (rank == 0) {
      memcpy(recv, send, bytes);
      for (int r = 1; r < world; r++)
        if (!recv_all(peers_[r], (char*)recv + (size_t)r * bytes, bytes)) return fail("rank " + std::to_string(r) + " went away");
      for (int r = 1; r < world; r++)
        if (!send_all(peers_[r], recv, bytes * (size_t)world)) return fail("rank " + std::to_string(r) + " went away");
      return true;
    }
    return (send_all(peers_[0], send, bytes) && recv_all(peers_[0], recv, bytes * (size_t)world)) || fail("rank 0 went away");
  }

  // buf of `root` reaches every rank
  bool broadcast(void* buf, size_t bytes, int root) {
    if (world == 1) return true;
    if (root < 0 || root >= world) return fail("bad root");
    if (rank == 0) {
      if (root != 0 && !recv_all(peers_[root], buf, bytes)) return fail("root went away");
      for (int r = 1; r < world; r++)
        if (r != root && !send_all(peers_[r], buf, bytes)) return fail("rank went away");
      return true;
    }
    if (rank == root) return send_all(peers_[0], buf, bytes) || fail("rank 0 went away");
    return recv_all(peers_[0], buf, bytes) || fail("rank 0 went away");
  }

  bool barrier() {
    std::vector<char> all((size_t)world);
    char one = 1;
    return all_gather(&one, 1, all.data());
  }

private:
  int listen_fd_ = -1;
  std::vector<int> peers_; // rank 0: peers_[r] = socket of rank r; others: peers_[0] = socket to rank 0

  bool fail(const std::string& m) {
    error = m;
    return false;
  }
  void close_all() {
    for (int fd : peers_)
      if (fd >= 0) ::close(fd);
    peers_.clear();
    if (listen_fd_ >= 0) ::close(listen_fd_);
    listen_fd_ = -1;
  }
  // no rank waits for ever on a peer that is alive but stuck: a transfer that makes no progress for
  // $PRV_COMM_TIMEOUT_S (default 600 s) fails, the call returns an error and the caller's error path runs
  static void tune(int fd) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    const char* e = getenv("PRV_COMM_TIMEOUT_S");
    const long secs = e && atol(e) > 0 ? atol(e) : 600;
    timeval tv{(time_t)secs, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
  }
  bool serve(const addrinfo* res, double timeout_s) {
    listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listen_fd_ < 0) return fail("socket()");
    int one = 1;
    setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in any = *(const sockaddr_in*)res->ai_addr;
    any.sin_addr.s_addr = htonl(INADDR_ANY);
    const auto t0 = std::chrono::steady_clock::now();
    while (::bind(listen_fd_, (const sockaddr*)&any, sizeof(any)) != 0) { // a previous job's listener may linger briefly
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("bind: " + std::string(strerror(errno)));
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (::listen(listen_fd_, world) != 0) return fail("listen()");
    timeval tv{(time_t)timeout_s, 0};
    setsockopt(listen_fd_, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    peers_.assign((size_t)world, -1);
    for (int k = 1; k < world; k++) {
      const int fd = ::accept(listen_fd_, nullptr, nullptr);
      if (fd < 0) return fail("accept: the other ranks did not arrive");
      tune(fd);
      int32_t hello[2] = {0, 0};
      if (!recv_all(fd, hello, sizeof(hello)) || hello[0] <= 0 || hello[0] >= world || hello[1] != world || peers_[hello[0]] >= 0) {
        ::close(fd);
        return fail("a peer introduced itself with a bad rank / world size");
      }
      peers_[hello[0]] = fd;
    }
    ::close(listen_fd_);
    listen_fd_ = -1;
    return true;
  }
  bool join(const addrinfo* res, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return fail("socket()");
      if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        tune(fd);
        const int32_t hello[2] = {rank, world};
        if (!send_all(fd, hello, sizeof(hello))) {
          ::close(fd);
          return fail("rank 0 closed the connection");
        }
        peers_.assign(1, fd);
        return true;
      }
      ::close(fd);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("connect: rank 0 is not listening");
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
};

} // namespace prvstar
