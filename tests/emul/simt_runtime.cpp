// SIMT emulation runtime — TEST INFRASTRUCTURE (see tests/emul/simt/hip/hip_runtime.h).
//
// simt::launch runs the blocks of a grid one after the other; the threads of a block are fibers with their own stacks,
// scheduled cooperatively from one OS thread.  A fiber runs until it finishes or reaches a cross-lane operation / barrier it
// cannot complete yet; the last participant to arrive completes the operation for everybody and makes the others runnable.
// A wave's participants are its lanes that have not returned yet (a returned lane contributes nothing, as on the hardware).
// Divergent use (lanes of one wave meeting in different kinds of operations, or nobody runnable while threads are still
// alive) aborts with a message: the product code is required to call the wave operations from wave-uniform control flow.
//
// Also here: the host versions of the two rocPRIM entry points of hgs_sort.h.
#include <hip/hip_runtime.h>

#include <stddef.h>

#include <algorithm>
#include <vector>

#include "../../hdl_graph_slam_amd/csrc/hgs_sort.h"

extern "C" void simt_switch(void** save_sp, void* new_sp);
// x86-64 SysV: callee-saved rbx, rbp, r12-r15 + the stack pointer make up a context
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size simt_switch,.-simt_switch
)");

namespace simt {

thread_local Thread* g_cur = nullptr;
unsigned long long g_record_fetches = 0, g_waves_launched = 0;
thread_local Idx g_block = {0, 0, 0}, g_block_dim = {1, 1, 1}, g_grid_dim = {1, 1, 1};

namespace {

enum State { READY = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
constexpr size_t kStackBytes = 256 * 1024;

struct Fiber {
  Thread th;
  State state = DONE;
  void* sp = nullptr;
  char* stack = nullptr;
};
struct Wave {
  int alive = 0, arrived = 0, kind = 0;
  unsigned gen = 0;
  unsigned long long live_mask = 0;
  unsigned long long vals[2][64];
  unsigned long long mask[2];
};

// one scheduler per host thread: engines are independent and may be driven from different threads
thread_local std::vector<Fiber> g_fibers;       // grown on demand, stacks reused between launches
thread_local std::vector<Wave> g_waves;
thread_local std::vector<int> g_ready;          // FIFO of runnable fibers
thread_local size_t g_ready_head = 0;
thread_local int g_nthreads = 0, g_block_alive = 0, g_block_arrived = 0;
thread_local void* g_sched_sp = nullptr;
thread_local const std::function<void()>* g_body = nullptr;

// HGS_SIMT_ORDER: 0 ascending (default), 1 reverse, 2 shuffle
int order_mode() {
  static const int mode = [] {
    const char* e = getenv("HGS_SIMT_ORDER");
    if (!e) return 0;
    if (!strncmp(e, "reverse", 7)) return 1;
    if (!strncmp(e, "shuffle", 7)) return 2;
    return 0;
  }();
  return mode;
}
unsigned next_random() {
  static thread_local unsigned long long state = [] {
    const char* e = getenv("HGS_SIMT_ORDER");
    const char* c = e ? strchr(e, ':') : nullptr;
    return 0x9E3779B97F4A7C15ull ^ (c ? strtoull(c + 1, nullptr, 10) : 1ull);
  }();
  state ^= state << 13, state ^= state >> 7, state ^= state << 17;
  return (unsigned)(state >> 16);
}

void make_ready(int f) {
  g_fibers[f].state = READY;
  g_ready.push_back(f);
}
inline Fiber& cur_fiber() { return *reinterpret_cast<Fiber*>(reinterpret_cast<char*>(g_cur) - offsetof(Fiber, th)); }
inline void yield_to_scheduler() {
  Fiber& me = cur_fiber();
  simt_switch(&me.sp, g_sched_sp);
}

// every live lane of the wave has arrived: publish who took part and release the waiting lanes
void complete_wave(Wave& w, int wave_index, int completer_lane /* -1: triggered by a lane that returned */) {
  unsigned long long took_part = completer_lane >= 0 ? 1ull << completer_lane : 0ull;
  for (int l = 0; l < 64; l++) {
    const int f = wave_index * 64 + l;
    if (f < g_nthreads && g_fibers[f].state == WAIT_WAVE) {
      took_part |= 1ull << l;
      make_ready(f);
    }
  }
  w.mask[w.gen & 1u] = took_part;
  w.arrived = 0;
  w.gen++;
}
void complete_block() {
  g_block_arrived = 0;
  for (int f = 0; f < g_nthreads; f++)
    if (g_fibers[f].state == WAIT_BLOCK) make_ready(f);
}

void fiber_main() {
  (*g_body)();
  Fiber& me = cur_fiber();
  me.state = DONE;
  Wave& w = g_waves[me.th.wave];
  w.alive--;
  w.live_mask &= ~(1ull << me.th.lane);
  g_block_alive--;
  // the lanes still waiting may have been waiting for this one only
  if (w.alive > 0 && w.arrived == w.alive) complete_wave(w, me.th.wave, -1);
  if (g_block_alive > 0 && g_block_arrived == g_block_alive) complete_block();
  simt_switch(&me.sp, g_sched_sp);
  fprintf(stderr, "simt: a finished fiber was resumed\n");
  abort();
}

void prepare(Fiber& f) {
  if (!f.stack) f.stack = (char*)aligned_alloc(64, kStackBytes);
  // top of stack: [.. r15 r14 r13 r12 rbx rbp | ret = fiber_main | fake return address]; fiber_main starts with rsp = 16n + 8
  uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
  void** p = (void**)top;
  *--p = nullptr;               // fake return address of fiber_main (never used)
  *--p = (void*)&fiber_main;    // popped by simt_switch's ret: this slot is 16-byte aligned
  for (int i = 0; i < 6; i++) *--p = nullptr;
  f.sp = p;
  f.state = READY;
}

}  // namespace

unsigned long long wave_exchange(int kind, unsigned long long mine, unsigned long long out[64]) {
  Fiber& me = cur_fiber();
  Wave& w = g_waves[me.th.wave];
  const unsigned gen = w.gen, buf = gen & 1u;
  if (w.arrived == 0) w.kind = kind;
  if (w.kind != kind) {
    fprintf(stderr, "simt: lanes of wave %d meet in different cross-lane operations (kind %d at source line %d vs kind %d at line %d): divergent "
            "control flow around a wave operation\n", me.th.wave, w.kind & 15, w.kind >> 4, kind & 15, kind >> 4);
    abort();
  }
  w.vals[buf][me.th.lane] = mine;
  w.arrived++;
  if (w.arrived == w.alive) {
    complete_wave(w, me.th.wave, me.th.lane);
  } else {
    me.state = WAIT_WAVE;
    yield_to_scheduler();
  }
  memcpy(out, w.vals[buf], sizeof(w.vals[buf]));
  return w.mask[buf];
}

void block_barrier() {
  Fiber& me = cur_fiber();
  g_block_arrived++;
  if (g_block_arrived == g_block_alive) {
    complete_block();
    return;
  }
  me.state = WAIT_BLOCK;
  yield_to_scheduler();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  if (g_cur) {
    fprintf(stderr, "simt: nested launch\n");
    abort();
  }
  const int nthreads = (int)(block.x * block.y * block.z);
  if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  const int nwaves = (nthreads + 63) / 64;
  g_waves.assign(nwaves, Wave{});
  g_nthreads = nthreads;
  g_body = &body;
  g_block_dim = {block.x, block.y, block.z};
  g_grid_dim = {grid.x, grid.y, grid.z};
  const bool backwards = order_mode() != 0;   // the blocks of a grid have no order either
  for (unsigned bzi = 0; bzi < grid.z; bzi++)
    for (unsigned byi = 0; byi < grid.y; byi++)
      for (unsigned bxi = 0; bxi < grid.x; bxi++) {
        const unsigned bx = backwards ? grid.x - 1 - bxi : bxi, by = backwards ? grid.y - 1 - byi : byi, bz = backwards ? grid.z - 1 - bzi : bzi;
        g_block = {bx, by, bz};
        g_ready.clear();
        g_ready_head = 0;
        for (int wv = 0; wv < nwaves; wv++) {
          Wave& w = g_waves[wv];
          w = Wave{};
          w.alive = std::min(64, nthreads - wv * 64);
          w.live_mask = w.alive == 64 ? ~0ull : ((1ull << w.alive) - 1);
        }
        g_block_alive = nthreads, g_block_arrived = 0;
        g_waves_launched += (unsigned long long)nwaves;
        for (int t = 0; t < nthreads; t++) {
          Fiber& f = g_fibers[t];
          f.th.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
          f.th.lane = t & 63, f.th.wave = t >> 6;
          prepare(f);
        }
        // start order of the fibers: ascending, or (HGS_SIMT_ORDER=reverse | shuffle[:seed]) another one — results must not
        // depend on which wave or lane runs first (a missing barrier or an order-dependent atomic shows up as a difference)
        for (int t = 0; t < nthreads; t++) g_ready.push_back(order_mode() == 1 ? nthreads - 1 - t : t);
        if (order_mode() == 2)
          for (int t = nthreads - 1; t > 0; t--) std::swap(g_ready[(size_t)t], g_ready[(size_t)(next_random() % (unsigned)(t + 1))]);
        while (g_block_alive > 0) {
          if (g_ready_head == g_ready.size()) {
            fprintf(stderr, "simt: deadlock in block (%u,%u,%u): %d threads alive, none runnable (a barrier or wave operation inside divergent code?)\n", bx,
                    by, bz, g_block_alive);
            abort();
          }
          const int f = g_ready[g_ready_head++];
          if (g_ready_head > 4096 && g_ready_head * 2 > g_ready.size()) {
            g_ready.erase(g_ready.begin(), g_ready.begin() + (long)g_ready_head);
            g_ready_head = 0;
          }
          if (g_fibers[f].state != READY) continue;
          g_cur = &g_fibers[f].th;
          simt_switch(&g_sched_sp, g_fibers[f].sp);
          g_cur = nullptr;
        }
      }
  g_body = nullptr;
}

}  // namespace simt

// ------------------------------------------------------------------------------------------------ hgs_sort.h on the host
extern "C" int hgs_sort_pairs_u64_u32(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                                      uint32_t* vals_out, size_t n, int begin_bit, int end_bit, void*) {
  if (!temp) {
    *temp_bytes = 256;
    return 0;
  }
  const uint64_t mask = (end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1)) & ~((1ull << begin_bit) - 1);
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
  std::vector<uint64_t> k(n);
  std::vector<uint32_t> v(n);
  for (size_t i = 0; i < n; i++) k[i] = keys_in[order[i]], v[i] = vals_in[order[i]];
  std::copy(k.begin(), k.end(), keys_out);
  std::copy(v.begin(), v.end(), vals_out);
  return 0;
}
extern "C" int hgs_exclusive_scan_u32(void* temp, size_t* temp_bytes, const uint32_t* in, uint32_t* out, size_t n, void*) {
  if (!temp) {
    *temp_bytes = 256;
    return 0;
  }
  uint32_t s = 0;
  for (size_t i = 0; i < n; i++) {
    const uint32_t v = in[i];
    out[i] = s;
    s += v;
  }
  return 0;
}

// statistics for scripts/walk_steps_emulated.py: {record fetches by lane 0 = packet-walk steps, waves launched}; reset on read
extern "C" void simt_read_counters(unsigned long long* out2) {
  out2[0] = simt::g_record_fetches, out2[1] = simt::g_waves_launched;
  simt::g_record_fetches = 0, simt::g_waves_launched = 0;
}

// ------------------------------------------------------------------------------------------------ hgs_comm.h on the host
// single-rank stand-in for the RCCL exchange step: the "all-gather" of one rank is a copy
#include "../../hdl_graph_slam_amd/csrc/hgs_comm.h"
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
// In-process stand-in for RCCL: the ranks of a communicator are engines driven from different host threads of ONE process (the
// emulated "devices" share the address space), meeting in a mutex / condition-variable rendezvous keyed by the unique id.  What it
// lets the CPU tests execute is the N > 1 path of hgs_loop_match_batch_sharded itself: headers, padding, merge, the failure paths.
namespace hgs {
namespace {
struct Group {
  std::mutex m;
  std::condition_variable cv;
  int world = 1, arrived = 0, departed = 0;
  long generation = 0;
  bool aborted = false;
  std::vector<const void*> send;
};
std::mutex g_groups_mutex;
std::map<std::string, std::shared_ptr<Group>> g_groups;
unsigned g_next_id = 1;
}  // namespace
struct Comm {
  int rank, world;
  std::shared_ptr<Group> group;
};
int comm_unique_id(void* id_out, char*, size_t) {
  std::lock_guard<std::mutex> lock(g_groups_mutex);
  memset(id_out, 0x5a, kCommUniqueIdBytes);
  const unsigned id = g_next_id++;
  memcpy(id_out, &id, sizeof(id));
  return 0;
}
int comm_create(Comm** out, int rank, int world, const void* id, int, char* err, size_t cap) {
  std::lock_guard<std::mutex> lock(g_groups_mutex);
  const std::string key((const char*)id, kCommUniqueIdBytes);
  std::shared_ptr<Group>& g = g_groups[key];
  if (!g) {
    g = std::make_shared<Group>();
    g->world = world;
    g->send.assign(world, nullptr);
  }
  if (g->world != world) {
    if (err && cap) snprintf(err, cap, "ranks disagree on the world size");
    return 1;
  }
  *out = new Comm{rank, world, g};
  return 0;
}
void comm_destroy(Comm* c) { delete c; }
void comm_abort(Comm* c) {
  if (!c || !c->group) return;
  std::lock_guard<std::mutex> lock(c->group->m);
  c->group->aborted = true;
  c->group->cv.notify_all();
}
int comm_async_error(Comm* c, char* err, size_t cap) {
  if (!c || !c->group) return 1;
  std::lock_guard<std::mutex> lock(c->group->m);
  if (c->group->aborted && err && cap) snprintf(err, cap, "the communicator has been aborted");
  return c->group->aborted ? 1 : 0;
}
int comm_rank(const Comm* c) { return c->rank; }
int comm_world(const Comm* c) { return c->world; }
int comm_all_gather(Comm* c, const void* send, void* recv, size_t bytes_per_rank, hipStream_t, char* err, size_t cap) {
  Group& g = *c->group;
  std::unique_lock<std::mutex> lock(g.m);
  auto aborted = [&]() {
    if (err && cap) snprintf(err, cap, "the communicator has been aborted");
    return 1;
  };
  // The real collective is enqueued and the engine waits for it with a deadline (comm_wait, HGS_COMM_TIMEOUT_MS); this stand-in completes inside the
  // call, so the same deadline applies to its rendezvous: a rank whose peers never arrive aborts the group and fails instead of blocking for ever.
  long limit_ms = 60000;
  if (const char* e = std::getenv("HGS_COMM_TIMEOUT_MS")) limit_ms = std::atol(e);
  auto wait_for = [&](auto pred) {
    if (limit_ms <= 0) {
      g.cv.wait(lock, pred);
      return true;
    }
    return g.cv.wait_for(lock, std::chrono::milliseconds(limit_ms), pred);
  };
  auto timed_out = [&]() {
    g.aborted = true;
    g.cv.notify_all();
    if (err && cap) snprintf(err, cap, "gave up waiting for the peers of the collective; the communicator has been aborted");
    return 1;
  };
  if (g.aborted) return aborted();
  // the previous collective's send buffers are released once everybody has left it
  if (!wait_for([&] { return g.departed == 0 || g.aborted; })) return timed_out();
  if (g.aborted) return aborted();
  g.send[c->rank] = send;
  const long gen = g.generation;
  if (++g.arrived == g.world) {
    g.arrived = 0, g.departed = g.world, g.generation++;
    g.cv.notify_all();
  } else {
    if (!wait_for([&] { return g.generation != gen || g.aborted; })) {
      g.arrived--;
      return timed_out();
    }
    if (g.generation == gen) return aborted();
  }
  for (int r = 0; r < g.world; r++) memmove((char*)recv + (size_t)r * bytes_per_rank, g.send[r], bytes_per_rank);
  // like the real collective, nobody's operation completes before its send buffer has been read by everyone (a rank may free it next)
  if (--g.departed == 0) g.cv.notify_all();
  else g.cv.wait(lock, [&] { return g.departed == 0 || g.aborted; });
  return 0;
}
}  // namespace hgs
