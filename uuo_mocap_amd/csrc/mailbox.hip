// Node-local exchange of the shared solves (uuo_lbfgs_solve_shared): a mailbox in POSIX shared memory.
//
// What crosses the ranks of a joint solve is 16 doubles per closure evaluation and 627 per iteration, produced by a kernel
// that already writes them into pinned HOST memory of its process (the report block the solver thread polls) and consumed by
// HOST code (the line search of every rank).  One process per GPU on one node: the shortest path from "rank r's report has
// arrived on r's host" to "every rank's host has it" is memory the hosts share.  Round 3 sent each block through a gloo
// all_gather (TCP over loopback, a Python callback under the interpreter lock: ~90 us per evaluation, as long as the
// evaluation itself); here rank r copies its block into its row of a shared table and reads the others' rows as their
// sequence words arrive -- no system call, no interpreter, no collective library on the path.  Ranks on different nodes
// (or an application that wants the collective on RCCL) keep the gather hook of dist_lbfgs.DistReducer.
//
// Protocol.  Gathers are numbered 1, 2, ... identically on every rank (the solver's decisions are identical everywhere, so
// all ranks issue the same sequence of gathers).  Row r has TWO slots; gather q uses slot q & 1.  Rank r writes {n,
// data} and then publishes q in the slot's sequence word (release); a reader waits for slot[q & 1].seq == q (acquire).  A
// rank can only write gather q + 2 after every rank has published q + 1, which each does only after it has READ all rows of
// q -- so a slot is never overwritten before everyone has read it.  A rank that fails still takes part in its gather (the
// driver's messages carry a status word, lbfgs_driver.hip shared_gather); a peer that died without a word is caught by the
// bounded wait.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "lbfgs.h"

#define UUO_MB_MAGIC 0x55554F4D41494C42ull  // "UUOMAILB"
#define UUO_MB_MAXN 640                     // doubles per message (Gram rows: 627 + status)

struct MbSlot {
  unsigned long long seq;  // the gather this slot holds (0 = never written)
  int n, pad;
  double data[UUO_MB_MAXN];
};
struct MbRow {
  MbSlot slot[2];
};
struct MbHeader {
  unsigned long long magic;
  int world, pad;
  unsigned long long opened;  // ranks that have mapped the table (rank 0 unlinks the name when all have)
};

struct uuo_mailbox {
  std::string name;
  int rank = 0, world = 1;
  bool owner = false;
  void* base = nullptr;
  size_t bytes = 0;
  unsigned long long seq = 0;
  unsigned long long total_ns = 0;  // time spent inside gathers (copy + waiting for the slowest rank)
  double timeout_s = 120.0;
  MbHeader* hdr() const { return reinterpret_cast<MbHeader*>(base); }
  MbRow* row(int r) const { return reinterpret_cast<MbRow*>((char*)base + 4096) + r; }
};

static size_t mb_bytes(int world) { return 4096 + (size_t)world * sizeof(MbRow); }

extern "C" int uuo_mailbox_close(uuo_mailbox_t* mb) {
  if (!mb) return 0;
  if (mb->base) munmap(mb->base, mb->bytes);
  if (mb->owner) shm_unlink(mb->name.c_str());  // (a no-op when every rank had opened it: unlinked then)
  delete mb;
  return 0;
}

extern "C" int uuo_mailbox_open(const char* name, int32_t rank, int32_t world, double timeout_s, uuo_mailbox_t** out) {
  UUO_REQUIRE(name && name[0] == '/' && out && world >= 1 && world <= 1024 && rank >= 0 && rank < world,
              "uuo_mailbox_open: bad arguments (the name must start with '/')");
  uuo_mailbox* mb = new uuo_mailbox();
  mb->name = name;
  mb->rank = rank;
  mb->world = world;
  mb->bytes = mb_bytes(world);
  if (timeout_s > 0) mb->timeout_s = timeout_s;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);  // a stale table of a crashed job with the same name
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)mb->bytes) != 0) {
      close(fd);
      shm_unlink(name);
      fd = -1;
    }
    mb->owner = fd >= 0;
  } else {
    UuoWaiter waiter;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {  // rank 0 may not have created it yet
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= mb->bytes) break;
      if (fd >= 0) close(fd);
      fd = -1;
      timespec ts{0, 200000};
      nanosleep(&ts, nullptr);
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) > mb->timeout_s) break;
    }
  }
  if (fd < 0) {
    uuo_set_error(std::string("uuo_mailbox_open: cannot open shared memory ") + name);
    delete mb;
    return -5;
  }
  mb->base = mmap(nullptr, mb->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (mb->base == MAP_FAILED) {
    mb->base = nullptr;
    uuo_set_error("uuo_mailbox_open: mmap failed");
    uuo_mailbox_close(mb);
    return -12;
  }
  MbHeader* h = mb->hdr();
  if (rank == 0) {  // fresh pages of a new object are zero: every sequence word starts at 0
    h->world = world;
    __atomic_store_n(&h->magic, UUO_MB_MAGIC, __ATOMIC_RELEASE);
  } else {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != UUO_MB_MAGIC) {
      timespec ts{0, 200000};
      nanosleep(&ts, nullptr);
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) > mb->timeout_s) {
        uuo_set_error("uuo_mailbox_open: rank 0 never initialised the table");
        uuo_mailbox_close(mb);
        return -62;
      }
    }
    if (h->world != world) {
      uuo_set_error("uuo_mailbox_open: the table was created for another world size");
      uuo_mailbox_close(mb);
      return -22;
    }
  }
  __atomic_add_fetch(&h->opened, 1ull, __ATOMIC_ACQ_REL);
  *out = mb;
  return 0;
}

// One gather: `mine[n]` of this rank into all[world][n], rank order.  A uuo_gather_fn (user = the mailbox): what
// uuo_shared_t.gather points at for the ranks of one node.
extern "C" int uuo_mailbox_gather(void* mailbox, const double* mine, int n, double* all) {
  uuo_mailbox* mb = reinterpret_cast<uuo_mailbox*>(mailbox);
  UUO_REQUIRE(mb && mine && all && n >= 0 && n <= UUO_MB_MAXN, "uuo_mailbox_gather: bad arguments / message too long");
  const unsigned long long q = ++mb->seq;
  MbSlot& me = mb->row(mb->rank)->slot[q & 1];
  me.n = n;
  std::memcpy(me.data, mine, sizeof(double) * (size_t)n);
  __atomic_store_n(&me.seq, q, __ATOMIC_RELEASE);
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < mb->world; ++r) {
    MbSlot& s = mb->row(r)->slot[q & 1];
    UuoWaiter waiter;
    while (__atomic_load_n(&s.seq, __ATOMIC_ACQUIRE) != q) {
      if (waiter.tick()) {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > mb->timeout_s) {
          uuo_set_error("shared solve: rank " + std::to_string(r) + " did not reach gather " + std::to_string(q) + " within " +
                        std::to_string((int)mb->timeout_s) + " s");
          return -62;
        }
      }
    }
    if (s.n != n) {
      uuo_set_error("shared solve: rank " + std::to_string(r) + " sent " + std::to_string(s.n) + " values where " +
                    std::to_string(n) + " were expected (the ranks are not in lock-step)");
      return -71;  // -EPROTO
    }
    std::memcpy(all + (size_t)r * n, s.data, sizeof(double) * (size_t)n);
  }
  timespec t2;
  clock_gettime(CLOCK_MONOTONIC, &t2);
  mb->total_ns += (unsigned long long)((t2.tv_sec - t0.tv_sec) * 1000000000ll + (t2.tv_nsec - t0.tv_nsec));
  return 0;
}

// gathers done through this mailbox so far and the time spent inside them (nanoseconds, including the wait for the slowest rank)
extern "C" int uuo_mailbox_stats(uuo_mailbox_t* mb, unsigned long long* gathers, unsigned long long* nanoseconds) {
  UUO_REQUIRE(mb && gathers && nanoseconds, "uuo_mailbox_stats: null argument");
  *gathers = mb->seq;
  *nanoseconds = mb->total_ns;
  return 0;
}
