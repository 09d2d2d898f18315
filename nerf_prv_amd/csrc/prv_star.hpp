// prv_star.hpp -- a tiny TCP star between the ranks of one job (POSIX sockets, host memory only).
//
// Two uses: (1) the bootstrap of the RCCL communicator in prv_comm.cpp (rank 0's ncclUniqueId reaches the
// other ranks through it -- what torch.distributed's TCP store does for the Python path), and (2) the
// `socket` transport shim of prv_comm: host-staged all-gather / broadcast for the cases RCCL does not cover
// (two ranks sharing one GPU in the tests, CPU-only rendezvous tests).  Not a data path for real multi-GPU
// runs: those gather device buffers with RCCL over xGMI.
//
// Topology: rank 0 listens on <addr>:<port> (MASTER_ADDR / MASTER_PORT + 23 by default, PRV_COMM_PORT
// overrides) -- bound to THAT address, not to every interface -- and every other rank connects (retrying until it
// is accepted).  A connection introduces itself with {magic, job nonce, communicator sequence number, rank, world};
// the nonce is a hash of MASTER_PORT and the job's token ($PRV_COMM_TOKEN, else $TORCHELASTIC_RUN_ID), the sequence
// number counts the communicators this process has opened.  Rank 0 closes and SKIPS anything that does not present
// the right nonce / sequence / a free rank (a port scanner, a straggler of another job, a peer's NEXT communicator
// landing in this listener's backlog) and keeps accepting until every rank has arrived or the timeout expires; a
// rank is accepted with an acknowledgement, and join() returns only once it has that, so a rejected connection is
// simply retried.  all_gather: every rank sends its block to rank 0, rank 0 answers with the assembled buffer.
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
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
  static constexpr uint32_t kMagic = 0x50525653u; // "PRVS"
  struct Hello {
    uint32_t magic;
    uint32_t seq;   // which communicator of the job this is (every rank opens them in the same order)
    uint64_t nonce; // the job: a hash of MASTER_PORT and the job token
    int32_t rank, world;
  };
  static uint64_t job_nonce() {
    uint64_t h = 0xcbf29ce484222325ull; // FNV-1a over the strings that name the job
    auto mix = [&](const char* s) {
      for (; s && *s; s++) h = (h ^ (uint64_t)(unsigned char)*s) * 0x100000001b3ull;
      h = (h ^ 0xffu) * 0x100000001b3ull;
    };
    mix(getenv("MASTER_PORT"));
    mix(getenv("PRV_COMM_TOKEN"));
    mix(getenv("TORCHELASTIC_RUN_ID"));
    return h;
  }
  static uint32_t next_seq() {
    static std::atomic<uint32_t> n{0};
    return n.fetch_add(1);
  }
  static double since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  static void set_timeouts(int fd, double secs) {
    if (secs < 0.05) secs = 0.05;
    timeval tv{(time_t)secs, (suseconds_t)((secs - (double)(time_t)secs) * 1e6)};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
  }
  bool serve(const addrinfo* res, double timeout_s) {
    const uint32_t seq = next_seq();
    const uint64_t nonce = job_nonce();
    listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listen_fd_ < 0) return fail("socket()");
    int one = 1;
    setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    const auto t0 = std::chrono::steady_clock::now();
    // the resolved rendezvous address itself (MASTER_ADDR), not INADDR_ANY: nothing outside that interface reaches the job
    while (::bind(listen_fd_, res->ai_addr, res->ai_addrlen) != 0) { // a previous job's listener may linger briefly
      if (since(t0) > timeout_s) return fail("bind: " + std::string(strerror(errno)));
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (::listen(listen_fd_, world + 16) != 0) return fail("listen()");
    peers_.assign((size_t)world, -1);
    int arrived = 1;
    while (arrived < world) {
      const double left = timeout_s - since(t0);
      if (left <= 0) return fail("accept: the other ranks did not arrive");
      set_timeouts(listen_fd_, left);
      const int fd = ::accept(listen_fd_, nullptr, nullptr);
      if (fd < 0) {
        if (errno == EINTR) continue;
        return fail("accept: the other ranks did not arrive");
      }
      set_timeouts(fd, std::min(left, 5.0)); // a connection that says nothing is dropped after 5 s, not waited on
      Hello h{};
      const bool good = recv_all(fd, &h, sizeof(h)) && h.magic == kMagic && h.nonce == nonce && h.seq == seq && h.world == world &&
                        h.rank > 0 && h.rank < world && peers_[(size_t)h.rank] < 0;
      if (!good) { // not one of this communicator's ranks: drop it, keep waiting for the real ones
        ::close(fd);
        continue;
      }
      const Hello ack{kMagic, seq, nonce, 0, world};
      if (!send_all(fd, &ack, sizeof(ack))) {
        ::close(fd);
        continue;
      }
      tune(fd);
      peers_[(size_t)h.rank] = fd;
      arrived++;
    }
    ::close(listen_fd_);
    listen_fd_ = -1;
    return true;
  }
  bool join(const addrinfo* res, double timeout_s) {
    const uint32_t seq = next_seq();
    const uint64_t nonce = job_nonce();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return fail("socket()");
      if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        set_timeouts(fd, std::max(1.0, std::min(timeout_s - since(t0), 30.0)));
        const Hello hello{kMagic, seq, nonce, rank, world};
        Hello ack{};
        // accepted = rank 0 answered with this communicator's acknowledgement; a listener that closes the connection
        // instead (it belongs to an earlier communicator or to another job) is retried
        if (send_all(fd, &hello, sizeof(hello)) && recv_all(fd, &ack, sizeof(ack)) && ack.magic == kMagic && ack.nonce == nonce &&
            ack.seq == seq && ack.world == world) {
          tune(fd);
          peers_.assign(1, fd);
          return true;
        }
      }
      ::close(fd);
      if (since(t0) > timeout_s) return fail("connect: rank 0 did not accept this rank");
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
};

} // namespace prvstar
