// Drives adapters/loop_match_hip.hpp the way a patched LoopDetector::matching would (loop_detector.hpp:117-171): one query
// keyframe, K candidate keyframes with their guesses, two detections in a row (the second reuses the resident keyframes).
//   loop_match_main <method> <n_engines> <target.bin> <guesses.bin> <cand0.bin> [cand1.bin ...]
// Clouds are raw pcl::PointXYZI records (32 bytes); guesses K x 16 floats column-major.  Prints, per detection:
//   best <index>
//   <candidate> <converged> <iterations> <fitness %.17g> <16 floats of final_transformation %.9g>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../adapters/loop_match_hip.hpp"

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
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
