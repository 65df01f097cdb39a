// loop_match_hip.hpp — header-only helper for the one caller that changes: LoopDetector::matching
// (include/hdl_graph_slam/loop_detector.hpp:117-171).  It owns
//   * one engine (hgs_handle) per GPU of the process — the ROS nodelet manager is a single process, so several GPUs are driven
//     from host threads instead of ranks (the per-candidate records are already in this process: no collective),
//   * the resident copies of the keyframe clouds (hgs_cloud), uploaded once per keyframe and reused by every later detection
//     together with their search index and covariances (keyframe k lives on engine k mod N: stable, balanced on average),
// and turns the sequential align / getFitnessScore loop of :135-154 into one hgs_loop_match_batch per engine.  The winner is
// chosen by hgs_select_best, i.e. by the reference's own rule (skip non-converged, skip score > best, ties replace).
// Only the C-ABI of include/hgs_registration.h and the standard library are used (no PCL types): the caller passes point
// arrays with a stride, e.g. cloud->points.data(), cloud->size(), sizeof(pcl::PointXYZI).
//
// Memory: a resident keyframe costs hgs_cloud_device_bytes() — about 100 bytes per point (12 MB for a 119 k-point HDL-64E sweep, 6 MB for a 60 k
// HDL-32E one: points, Hilbert-sorted copy, leaf copy, box tree, covariances, correspondence seeds).  hdl_graph_slam never removes a keyframe from its
// graph, so the resident set is BOUNDED here: setCapacity(bytes per engine, keyframes per engine) — default 16 GiB / unlimited count of the 288 GB —
// and after every match() the least recently matched keyframes beyond the budget are released (a keyframe that becomes a candidate again is simply
// uploaded again: the host copy is the graph's KeyFrame::cloud).  The candidates of the running batch are never evicted under it.
//
//   hgs_hip::LoopMatcherHIP matcher(params, {0, 1, 2, 3});
//   std::vector<hgs_hip::LoopMatcherHIP::Candidate> cands;          // {keyframe id, points, n, stride, guess[16]}
//   ...
//   int best = matcher.match(new_keyframe_points, n, stride, cands, fitness_score_max_range, &records);
//
// Exercised end to end by tests/cpp/loop_match_main.cpp (tests/test_simt_kernels_host.py) on the host emulation of the kernels.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "hgs_registration.h"

namespace hgs_hip {

class LoopMatcherHIP {
public:
  struct Candidate {
    long keyframe_id;       // identifies the resident copy (KeyFrame::id())
    const void* points;     // only read when the keyframe is not resident yet
    size_t n;
    size_t stride_bytes;
    float guess[16];        // column-major initial guess (loop_detector.hpp:137-142)
  };

  LoopMatcherHIP(const hgs_params& params, const std::vector<int>& device_ids) {
    if (device_ids.empty()) throw std::invalid_argument("LoopMatcherHIP: no device");
    for (int d : device_ids) {
      hgs_params p = params;
      p.device_id = d;
      hgs_handle* h = nullptr;
      const int rc = hgs_create(&p, &h);
      if (rc != HGS_OK) {
        const std::string msg = hgs_last_error(nullptr);
        release();
        throw std::runtime_error("LoopMatcherHIP: hgs_create failed: " + msg);
      }
      engines_.push_back(Engine{h, {}, 0, 0});
    }
  }
  ~LoopMatcherHIP() { release(); }
  LoopMatcherHIP(const LoopMatcherHIP&) = delete;
  LoopMatcherHIP& operator=(const LoopMatcherHIP&) = delete;

  size_t num_engines() const { return engines_.size(); }
  size_t resident_keyframes() const {
    size_t n = 0;
    for (const Engine& e : engines_) n += e.clouds.size();
    return n;
  }
  // device bytes held by the resident keyframes of all engines (hgs_cloud_device_bytes, refreshed after every match)
  size_t resident_bytes() const {
    size_t n = 0;
    for (const Engine& e : engines_) n += e.bytes;
    return n;
  }
  size_t evictions() const {
    size_t n = 0;
    for (const Engine& e : engines_) n += e.evictions;
    return n;
  }
  // Budget PER ENGINE (= per GPU); 0 = unlimited.  Takes effect at the end of the next match().
  void setCapacity(size_t max_bytes_per_engine, size_t max_keyframes_per_engine = 0) { max_bytes_ = max_bytes_per_engine, max_keyframes_ = max_keyframes_per_engine; }
  // a keyframe was removed from the graph / its cloud changed
  void forget(long keyframe_id) {
    Engine& e = owner(keyframe_id);
    auto it = e.clouds.find(keyframe_id);
    if (it == e.clouds.end()) return;
    drop(e, it);
  }
  // every resident keyframe (e.g. the graph was replaced by load_service: apps/hdl_graph_slam_nodelet.cpp:932-974 reuses node ids)
  void forget_all() {
    for (Engine& e : engines_)
      while (!e.clouds.empty()) drop(e, e.clouds.begin());
  }

  // Registers every candidate against the new keyframe; records[i] is filled for candidates[i].  Returns the index the
  // sequential rule of loop_detector.hpp:146-153 selects, or -1.  Never throws into the caller (LoopDetector::matching has no
  // handler and the nodelet would die): when an engine fails, ITS candidates stay "not converged" (fitness DBL_MAX — the rule
  // skips them, like the sharded C entry point does for a rank that failed), the other engines' results stand, and
  // last_error() says what happened.
  int match(const void* target_points, size_t target_n, size_t target_stride, const std::vector<Candidate>& candidates, double max_range,
            std::vector<hgs_result>* records) {
    const size_t N = engines_.size();
    const uint64_t tick = ++tick_;
    std::vector<std::vector<size_t>> mine(N);
    for (size_t i = 0; i < candidates.size(); i++) mine[owner_index(candidates[i].keyframe_id)].push_back(i);
    records->assign(candidates.size(), hgs_result{});
    for (size_t i = 0; i < candidates.size(); i++) {
      (*records)[i].candidate_id = (int32_t)i;
      (*records)[i].fitness_score = 1.7976931348623157e308;  // DBL_MAX: not converged until an engine says otherwise
    }
    last_error_.clear();
    std::vector<std::string> errors(N);
    auto work = [&](size_t e) {
      try {
        Engine& eng = engines_[e];
        if (mine[e].empty()) return;
        check(eng, hgs_set_target(eng.h, target_points, target_n, target_stride));   // replicated: cheaper than a broadcast
        std::vector<hgs_cloud*> clouds;
        std::vector<float> guesses;
        for (size_t i : mine[e]) {
          const Candidate& c = candidates[i];
          auto it = eng.clouds.find(c.keyframe_id);
          if (it == eng.clouds.end()) {
            hgs_cloud* cl = nullptr;
            check(eng, hgs_cloud_create(eng.h, c.points, c.n, c.stride_bytes, &cl));
            it = eng.clouds.emplace(c.keyframe_id, Resident{cl, 0, 0}).first;
          }
          it->second.last_used = tick;
          clouds.push_back(it->second.cloud);
          guesses.insert(guesses.end(), c.guess, c.guess + 16);
        }
        std::vector<hgs_result> out(clouds.size());
        int32_t best_local = -1;
        check(eng, hgs_loop_match_batch(eng.h, clouds.data(), clouds.size(), guesses.data(), max_range, out.data(), &best_local));
        for (size_t k = 0; k < out.size(); k++) {
          out[k].candidate_id = (int32_t)mine[e][k];
          (*records)[mine[e][k]] = out[k];
        }
      } catch (const std::exception& ex) {
        errors[e] = ex.what();
      }
      try {
        enforce_capacity(engines_[e], tick);  // (also after a failure: a failed hgs_cloud_create is most likely an exhausted device)
      } catch (const std::exception& ex) {
        if (errors[e].empty()) errors[e] = ex.what();
      }
    };
    if (N == 1) {
      work(0);
    } else {
      std::vector<std::thread> threads;
      for (size_t e = 0; e < N; e++) threads.emplace_back(work, e);
      for (std::thread& t : threads) t.join();
    }
    failed_engines_ = 0, used_engines_ = 0;
    for (size_t e = 0; e < N; e++) {
      if (!mine[e].empty()) used_engines_++;
      if (!errors[e].empty()) last_error_ += "engine " + std::to_string(e) + ": " + errors[e] + "; ", failed_engines_++;
    }
    int32_t best = -1;
    if (!candidates.empty() && hgs_select_best(records->data(), records->size(), &best) != HGS_OK) last_error_ += "hgs_select_best failed; ", best = -1;
    return best;
  }
  // empty after a clean match(); otherwise which engine failed and why (its candidates were left not converged)
  const std::string& last_error() const { return last_error_; }
  // engines of the last match() that had candidates / that failed: a caller that must not lose candidates silently (LoopDetector::matching)
  // falls back to its sequential loop when failed_engines() > 0
  size_t used_engines() const { return used_engines_; }
  size_t failed_engines() const { return failed_engines_; }

private:
  struct Resident {
    hgs_cloud* cloud;
    uint64_t last_used;   // match() counter of the last batch this keyframe took part in
    size_t bytes;         // hgs_cloud_device_bytes at the end of that batch (index / covariances are built inside it)
  };
  struct Engine {
    hgs_handle* h;
    std::unordered_map<long, Resident> clouds;
    size_t bytes = 0;
    size_t evictions = 0;  // (per engine: the engines of a match() run on their own threads)
  };
  std::vector<Engine> engines_;
  std::string last_error_;
  size_t max_bytes_ = (size_t)16 << 30, max_keyframes_ = 0, used_engines_ = 0, failed_engines_ = 0;
  uint64_t tick_ = 0;

  void drop(Engine& e, std::unordered_map<long, Resident>::iterator it) {
    e.bytes -= std::min(e.bytes, it->second.bytes);
    hgs_cloud_destroy(it->second.cloud);
    e.clouds.erase(it);
  }
  // Called by the engine's own worker thread at the end of a batch: refresh the sizes of the keyframes the batch touched, then release the least
  // recently matched ones until the engine is within its budget.  Keyframes of the current batch (last_used == tick) go last and only if the
  // batch alone exceeds the budget.
  void enforce_capacity(Engine& e, uint64_t tick) {
    for (auto& kv : e.clouds) {
      if (kv.second.last_used != tick) continue;
      const size_t b = hgs_cloud_device_bytes(kv.second.cloud);
      e.bytes = e.bytes - std::min(e.bytes, kv.second.bytes) + b;
      kv.second.bytes = b;
    }
    auto over = [&]() { return (max_bytes_ && e.bytes > max_bytes_) || (max_keyframes_ && e.clouds.size() > max_keyframes_); };
    if (!over()) return;
    std::vector<std::pair<uint64_t, long>> order;
    order.reserve(e.clouds.size());
    for (const auto& kv : e.clouds) order.emplace_back(kv.second.last_used, kv.first);
    std::sort(order.begin(), order.end());
    for (const auto& o : order) {
      if (!over()) break;
      drop(e, e.clouds.find(o.second));
      e.evictions++;
    }
  }

  size_t owner_index(long keyframe_id) const { return (size_t)((keyframe_id % (long)engines_.size() + (long)engines_.size()) % (long)engines_.size()); }
  Engine& owner(long keyframe_id) { return engines_[owner_index(keyframe_id)]; }
  static void check(Engine& e, int rc) {
    if (rc != HGS_OK) throw std::runtime_error(std::string("hgs: ") + hgs_last_error(e.h));
  }
  void release() {
    for (Engine& e : engines_) {
      for (auto& kv : e.clouds) hgs_cloud_destroy(kv.second.cloud);
      e.clouds.clear();
      e.bytes = 0;
      hgs_destroy(e.h);
    }
    engines_.clear();
  }
};

}  // namespace hgs_hip
