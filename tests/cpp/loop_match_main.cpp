// Drives adapters/loop_match_hip.hpp the way a patched LoopDetector::matching would (loop_detector.hpp:117-171): one query
// keyframe, K candidate keyframes with their guesses, two detections in a row (the second reuses the resident keyframes).
//   loop_match_main <method> <n_engines> <target.bin> <guesses.bin> <cand0.bin> [cand1.bin ...]
// Clouds are raw pcl::PointXYZI records (32 bytes); guesses K x 16 floats column-major.  Prints, per detection:
//   best <index>
//   <candidate> <converged> <iterations> <fitness %.17g> <16 floats of final_transformation %.9g>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <algorithm>
#include <cstring>
#include <iterator>
#include <string>
#include <vector>

#include "../../adapters/loop_match_hip.hpp"

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// loop_match_main capacity <method> <n_keyframes> <keyframes per detection> <max resident keyframes> <max resident MiB> <points per keyframe>
// A long run of detections over a growing pose graph (hdl_graph_slam never removes a keyframe): synthetic keyframes (a noisy plane + two walls, shifted
// per keyframe), every detection matches the next few keyframe ids plus one old one against a target; prints the high-water marks of the resident set and
// checks that a keyframe that was evicted and uploaded again gives the SAME record as the first time.
static int run_capacity(int argc, char** argv) {
  if (argc < 8) return 2;
  const int method = std::atoi(argv[2]), n_kf = std::atoi(argv[3]), per = std::atoi(argv[4]), max_kf = std::atoi(argv[5]);
  const double max_mib = std::atof(argv[6]);
  const int npts = std::atoi(argv[7]);
  hgs_params p;
  if (hgs_params_default(method, &p) != HGS_OK) return 3;
  if (method == HGS_NDT_OMP) p.resolution = 1.0;
  struct Pt {
    float x, y, z, w, intensity, pad[3];
  };
  unsigned long long rng = 12345;
  auto uni = [&]() {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((rng >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
  };
  auto make_cloud = [&](double shift) {
    std::vector<Pt> c((size_t)npts);
    for (int i = 0; i < npts; i++) {
      const double u = uni() * 20 - 10, v = uni() * 20 - 10, n = (uni() - 0.5) * 0.04;
      Pt q{};
      if (i % 3 == 0) q.x = (float)(u + shift), q.y = (float)v, q.z = (float)n;            // ground
      else if (i % 3 == 1) q.x = (float)(6.0 + n + shift), q.y = (float)v, q.z = (float)(uni() * 3);  // wall across x
      else q.x = (float)(u + shift), q.y = (float)(-7.0 + n), q.z = (float)(uni() * 3);   // wall along x
      q.w = 1.f;
      c[(size_t)i] = q;
    }
    return c;
  };
  const std::vector<Pt> target = make_cloud(0.0);
  std::vector<std::vector<Pt>> kfs;
  for (int k = 0; k < n_kf; k++) kfs.push_back(make_cloud(0.05 * ((k * 7) % 11 - 5)));
  try {
    hgs_hip::LoopMatcherHIP matcher(p, std::vector<int>{0});
    matcher.setCapacity((size_t)(max_mib * 1024.0 * 1024.0), (size_t)max_kf);
    size_t hw_kf = 0, hw_bytes = 0, detections = 0, mismatches = 0;
    std::vector<hgs_result> first((size_t)n_kf);
    std::vector<char> seen((size_t)n_kf, 0);
    for (int k0 = 0; k0 < n_kf; k0 += per) {
      std::vector<hgs_hip::LoopMatcherHIP::Candidate> cands;
      std::vector<int> ids;
      for (int k = k0; k < std::min(n_kf, k0 + per); k++) ids.push_back(k);
      if (k0 >= 8 * per) ids.push_back((int)((k0 * 5) % (k0 - 4 * per)));  // an old keyframe, long evicted under a small budget
      for (int id : ids) {
        hgs_hip::LoopMatcherHIP::Candidate cd;
        cd.keyframe_id = id, cd.points = kfs[(size_t)id].data(), cd.n = kfs[(size_t)id].size(), cd.stride_bytes = sizeof(Pt);
        const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::memcpy(cd.guess, I, sizeof(I));
        cands.push_back(cd);
      }
      std::vector<hgs_result> rec;
      matcher.match(target.data(), target.size(), sizeof(Pt), cands, 4.0, &rec);
      if (!matcher.last_error().empty()) {
        std::fprintf(stderr, "error: %s\n", matcher.last_error().c_str());
        return 1;
      }
      detections++;
      hw_kf = std::max(hw_kf, matcher.resident_keyframes()), hw_bytes = std::max(hw_bytes, matcher.resident_bytes());
      for (size_t i = 0; i < ids.size(); i++) {
        hgs_result r = rec[i];
        r.candidate_id = 0;
        if (!seen[(size_t)ids[i]]) first[(size_t)ids[i]] = r, seen[(size_t)ids[i]] = 1;
        else if (std::memcmp(&first[(size_t)ids[i]], &r, sizeof(r)) != 0) mismatches++;
      }
    }
    std::printf("capacity detections %zu keyframes %d high_water_keyframes %zu high_water_bytes %zu evictions %zu resident_now %zu mismatches %zu\n", detections, n_kf, hw_kf,
                hw_bytes, matcher.evictions(), matcher.resident_keyframes(), mismatches);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "capacity") return run_capacity(argc, argv);
  if (argc < 6) return 2;
  const int method = std::atoi(argv[1]), n_engines = std::atoi(argv[2]);
  hgs_params p;
  if (hgs_params_default(method, &p) != HGS_OK) return 3;
  if (method == HGS_NDT_OMP) p.resolution = 1.0;
  const std::vector<char> target = slurp(argv[3]), guesses = slurp(argv[4]);
  std::vector<std::vector<char>> clouds;
  for (int i = 5; i < argc; i++) clouds.push_back(slurp(argv[i]));
  const size_t K = clouds.size(), stride = 32;
  try {
    hgs_hip::LoopMatcherHIP matcher(p, std::vector<int>((size_t)n_engines, 0));   // every engine on device 0: what a 1-GPU box can run
    for (int detection = 0; detection < 2; detection++) {
      std::vector<hgs_hip::LoopMatcherHIP::Candidate> cands;
      for (size_t i = 0; i < K; i++) {
        const size_t c = detection == 0 ? i : K - 1 - i;   // second detection: same keyframes, other order
        if (detection == 1 && c == 0) continue;            // ... and one fewer
        hgs_hip::LoopMatcherHIP::Candidate cd;
        cd.keyframe_id = 100 + (long)c, cd.points = clouds[c].data(), cd.n = clouds[c].size() / stride, cd.stride_bytes = stride;
        std::memcpy(cd.guess, guesses.data() + c * 16 * sizeof(float), 16 * sizeof(float));
        cands.push_back(cd);
      }
      std::vector<hgs_result> rec;
      const int best = matcher.match(target.data(), target.size() / stride, stride, cands, 4.0, &rec);
      std::printf("best %d resident %zu\n", best, matcher.resident_keyframes());
      for (size_t i = 0; i < rec.size(); i++) {
        std::printf("%ld %d %d %.17g", cands[i].keyframe_id - 100, rec[i].converged, rec[i].iterations, rec[i].fitness_score);
        for (int k = 0; k < 16; k++) std::printf(" %.9g", rec[i].final_transformation[k]);
        std::printf("\n");
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
