// prv_star.hpp -- a tiny TCP star between the ranks of one job (POSIX sockets, host memory only).
//
// Two uses: (1) the bootstrap of the RCCL communicator in prv_comm.cpp (rank 0's ncclUniqueId reaches the
// other ranks through it -- what torch.distributed's TCP store does for the Python path), and (2) the
// `socket` transport shim of prv_comm: host-staged all-gather / broadcast for the cases RCCL does not cover
// (two ranks sharing one GPU in the tests, CPU-only rendezvous tests).  Not a data path for real multi-GPU
// runs: those gather device buffers with RCCL over xGMI.
//
// Topology: rank 0 listens on <port> (MASTER_PORT + 23 by default, PRV_COMM_PORT overrides) and every other rank connects
// to <addr>:<port> (MASTER_ADDR), retrying until it is accepted.  Rank 0 binds the rendezvous address itself when that
// is a literal IP of a local interface (nothing outside that interface reaches the job); when MASTER_ADDR is a NAME
// that resolves to loopback on this host (Debian's 127.0.1.1 for the own hostname) or an address that is not local
// (a service / NAT address: EADDRNOTAVAIL) it listens on every interface instead -- remote ranks must be able to reach
// it, and the hello below keeps strangers out.  A bind error other than "address in use" is reported at once.
// A connection introduces itself with {magic, job nonce, rank, world}; the nonce is a hash of MASTER_PORT and the job's
// token ($PRV_COMM_TOKEN, else $TORCHELASTIC_RUN_ID).  Rank 0 closes without a word anything that does not present the
// magic and the nonce (a port scanner, another job), REFUSES with a reason a rank of this job that cannot be seated
// (another world size: fatal for the joiner; a rank already seated -- a peer's NEXT communicator landing in this
// listener's backlog: the joiner retries), and seats everything else with an acknowledgement that carries rank 0's
// communicator sequence number: rank 0 dictates it, so a rank whose earlier prv_comm_create failed before it got here
// is not out of step for the rest of the job.  join() returns once it is seated.
// all_gather: every rank sends its block to rank 0, rank 0 answers with the assembled buffer.
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
#include <cstdio>
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
    bool ok = rank == 0 ? serve(res, timeout_s, addr) : join(res, timeout_s);
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
  enum : uint32_t { kSeated = 0, kRetry = 1, kWorldMismatch = 2, kBadRank = 3 };
  struct Hello {
    uint32_t magic;
    uint32_t seq;   // joiner -> rank 0: how many stars this process has opened before this one; rank 0 -> joiner: the communicator's sequence number at rank 0
    uint64_t nonce; // the job: a hash of MASTER_PORT and the job token
    int32_t rank, world;
    uint32_t verdict, pad; // rank 0's answer: kSeated, or why not
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
  static bool job_has_token() {
    const char* a = getenv("PRV_COMM_TOKEN");
    const char* b = getenv("TORCHELASTIC_RUN_ID");
    return (a && *a) || (b && *b);
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
public:
  uint32_t seq = 0; // this communicator's sequence number as rank 0 counts them (diagnostics)

private:
  // rank 0's listening address: see the header comment
  static bool is_ip_literal(const std::string& a) {
    in_addr tmp;
    return inet_pton(AF_INET, a.c_str(), &tmp) == 1;
  }
  bool serve(const addrinfo* res, double timeout_s, const std::string& addr_text) {
    seq = next_seq();
    const uint64_t nonce = job_nonce();
    listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listen_fd_ < 0) return fail("socket()");
    int one = 1;
    setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    const auto t0 = std::chrono::steady_clock::now();
    sockaddr_in want{};
    memcpy(&want, res->ai_addr, std::min(sizeof(want), (size_t)res->ai_addrlen));
    const bool loopback = (ntohl(want.sin_addr.s_addr) >> 24) == 127;
    if (loopback && !is_ip_literal(addr_text) && addr_text != "localhost") want.sin_addr.s_addr = htonl(INADDR_ANY);
    for (;;) {
      if (::bind(listen_fd_, (const sockaddr*)&want, sizeof(want)) == 0) break;
      const int err = errno;
      if (err == EADDRNOTAVAIL && want.sin_addr.s_addr != htonl(INADDR_ANY)) { // not an address of this host: every interface
        want.sin_addr.s_addr = htonl(INADDR_ANY);
        continue;
      }
      if (err != EADDRINUSE) return fail("bind " + addr_text + ": " + std::string(strerror(err))); // nothing waiting will cure
      if (since(t0) > timeout_s) return fail("bind " + addr_text + ": " + std::string(strerror(err)) + " (for the whole timeout)");
      std::this_thread::sleep_for(std::chrono::milliseconds(100)); // a previous job's listener may linger briefly
    }
    if (want.sin_addr.s_addr == htonl(INADDR_ANY) && !job_has_token()) {
      // every interface, and the only thing that tells this job's ranks from a stranger is a hash of MASTER_PORT: say so once
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true))
        fprintf(stderr, "prv_star: rank 0 listens on EVERY interface (MASTER_ADDR '%s' is not a local literal) and the job has no token: "
                        "set PRV_COMM_TOKEN (or run under torchrun, TORCHELASTIC_RUN_ID) so that only this job's ranks are seated\n", addr_text.c_str());
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
      if (!recv_all(fd, &h, sizeof(h)) || h.magic != kMagic || h.nonce != nonce) { // a stranger: not a word
        ::close(fd);
        continue;
      }
      Hello ack{kMagic, seq, nonce, 0, world, kSeated, 0};
      if (h.world != world) ack.verdict = kWorldMismatch;
      else if (h.rank <= 0 || h.rank >= world) ack.verdict = kBadRank;
      else if (peers_[(size_t)h.rank] >= 0) ack.verdict = kRetry; // that rank sits here already: its NEXT communicator, too early
      if (!send_all(fd, &ack, sizeof(ack)) || ack.verdict != kSeated) {
        ::close(fd);
        continue;
      }
      if (h.seq != seq) // seated all the same (rank 0 dictates the number), but a straggler of an attempt this rank gave up on looks like this
        fprintf(stderr, "prv_star: rank %d joins communicator #%u of rank 0 as its own #%u: the ranks do not count alike (an earlier rendezvous "
                        "timed out on one side?)\n", h.rank, seq, h.seq);
      tune(fd);
      peers_[(size_t)h.rank] = fd;
      arrived++;
    }
    ::close(listen_fd_);
    listen_fd_ = -1;
    return true;
  }
  bool join(const addrinfo* res, double timeout_s) {
    const uint64_t nonce = job_nonce();
    const uint32_t mine = next_seq(); // this process's own count of stars (rank 0 compares, prv_star: ... do not count alike)
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return fail("socket()");
      if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        set_timeouts(fd, std::max(1.0, std::min(timeout_s - since(t0), 30.0)));
        const Hello hello{kMagic, mine, nonce, rank, world, 0, 0};
        Hello ack{};
        // seated = rank 0 answered with an acknowledgement; a listener that closes the connection without a word is
        // not this job's (or died), a kRetry answer is this job's listener of an earlier communicator: try again
        if (send_all(fd, &hello, sizeof(hello)) && recv_all(fd, &ack, sizeof(ack)) && ack.magic == kMagic && ack.nonce == nonce) {
          if (ack.verdict == kSeated) {
            seq = ack.seq;
            tune(fd);
            peers_.assign(1, fd);
            return true;
          }
          if (ack.verdict == kWorldMismatch || ack.verdict == kBadRank) {
            ::close(fd);
            return fail(ack.verdict == kWorldMismatch
                            ? "rank 0 refused this rank: it runs a communicator of " + std::to_string(ack.world) + " ranks, this rank expects " + std::to_string(world)
                            : "rank 0 refused this rank: rank " + std::to_string(rank) + " is outside its world of " + std::to_string(ack.world));
          }
        }
      }
      ::close(fd);
      if (since(t0) > timeout_s) return fail("connect: rank 0 did not seat this rank within the timeout");
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
};

} // namespace prvstar
