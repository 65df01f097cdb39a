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
//   hgs_hip::LoopMatcherHIP matcher(params, {0, 1, 2, 3});
//   std::vector<hgs_hip::LoopMatcherHIP::Candidate> cands;          // {keyframe id, points, n, stride, guess[16]}
//   ...
//   int best = matcher.match(new_keyframe_points, n, stride, cands, fitness_score_max_range, &records);
//
// Exercised end to end by tests/cpp/loop_match_main.cpp (tests/test_simt_kernels_host.py) on the host emulation of the kernels.
#pragma once

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
      engines_.push_back(Engine{h, {}});
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
  // a keyframe was removed from the graph / its cloud changed
  void forget(long keyframe_id) {
    Engine& e = owner(keyframe_id);
    auto it = e.clouds.find(keyframe_id);
    if (it == e.clouds.end()) return;
    hgs_cloud_destroy(it->second);
    e.clouds.erase(it);
  }

  // Registers every candidate against the new keyframe; records[i] is filled for candidates[i].  Returns the index the
  // sequential rule of loop_detector.hpp:146-153 selects, or -1.  Never throws into the caller (LoopDetector::matching has no
  // handler and the nodelet would die): when an engine fails, ITS candidates stay "not converged" (fitness DBL_MAX — the rule
  // skips them, like the sharded C entry point does for a rank that failed), the other engines' results stand, and
  // last_error() says what happened.
  int match(const void* target_points, size_t target_n, size_t target_stride, const std::vector<Candidate>& candidates, double max_range,
            std::vector<hgs_result>* records) {
    const size_t N = engines_.size();
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
            it = eng.clouds.emplace(c.keyframe_id, cl).first;
          }
          clouds.push_back(it->second);
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
    };
    if (N == 1) {
      work(0);
    } else {
      std::vector<std::thread> threads;
      for (size_t e = 0; e < N; e++) threads.emplace_back(work, e);
      for (std::thread& t : threads) t.join();
    }
    for (size_t e = 0; e < N; e++)
      if (!errors[e].empty()) last_error_ += "engine " + std::to_string(e) + ": " + errors[e] + "; ";
    int32_t best = -1;
    if (!candidates.empty() && hgs_select_best(records->data(), records->size(), &best) != HGS_OK) last_error_ += "hgs_select_best failed; ", best = -1;
    return best;
  }
  // empty after a clean match(); otherwise which engine failed and why (its candidates were left not converged)
  const std::string& last_error() const { return last_error_; }

private:
  struct Engine {
    hgs_handle* h;
    std::unordered_map<long, hgs_cloud*> clouds;
  };
  std::vector<Engine> engines_;
  std::string last_error_;

  size_t owner_index(long keyframe_id) const { return (size_t)((keyframe_id % (long)engines_.size() + (long)engines_.size()) % (long)engines_.size()); }
  Engine& owner(long keyframe_id) { return engines_[owner_index(keyframe_id)]; }
  static void check(Engine& e, int rc) {
    if (rc != HGS_OK) throw std::runtime_error(std::string("hgs: ") + hgs_last_error(e.h));
  }
  void release() {
    for (Engine& e : engines_) {
      for (auto& kv : e.clouds) hgs_cloud_destroy(kv.second);
      e.clouds.clear();
      hgs_destroy(e.h);
    }
    engines_.clear();
  }
};

}  // namespace hgs_hip
