// batch_multi_gpu — BASELINE.json config 5 from a C++ host: a batch of independent stereo pairs sharded over the GPUs of
// a node (one context + one host thread per GPU), one RCCL gather of the result records (include/bpvo_hip/multi_gpu.h).
//
//   batch_multi_gpu <dir> <rows> <cols> <fx> <fy> <cx> <cy> <baseline> <n_pairs> <n_gpus> [intensity|bitplanes] [levels] [output_prefix]
//
// <dir> holds images.u8 (2*n_pairs images A0,B0,A1,B1,... of rows*cols bytes) and disparities.f32 (likewise, floats) — the
// layout bpvo_hip_batch_run takes.  Writes <prefix>_poses.f32 (n_pairs x 16) and <prefix>_records.f32 (n_pairs x 32) and
// prints the wall time of the sharded run (host buffers, PCIe included).
#include <bpvo_hip/multi_gpu.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

template <typename T>
static bool readRaw(const std::string& path, std::vector<T>& out, size_t n)
{
  std::ifstream f(path, std::ios::binary);
  if(!f) return false;
  out.resize(n);
  f.read(reinterpret_cast<char*>(out.data()), n * sizeof(T));
  return (size_t) f.gcount() == n * sizeof(T);
}

static bool writeRaw(const std::string& path, const std::vector<float>& v)
{
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(float));
  return (bool) f;
}

int main(int argc, char** argv)
{
  if(argc < 11) {
    std::fprintf(stderr, "usage: %s dir rows cols fx fy cx cy baseline n_pairs n_gpus [intensity|bitplanes] [levels] [output_prefix]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
  const float K[9] = {(float) std::atof(argv[4]), 0.0f, (float) std::atof(argv[6]), 0.0f, (float) std::atof(argv[5]), (float) std::atof(argv[7]), 0.0f, 0.0f, 1.0f};
  const float baseline = (float) std::atof(argv[8]);
  const int n_pairs = std::atoi(argv[9]), n_gpus = std::atoi(argv[10]);
  const std::string desc = argc > 11 ? argv[11] : "bitplanes";
  const int levels = argc > 12 ? std::atoi(argv[12]) : 4;
  const std::string prefix = argc > 13 ? argv[13] : "";

  bpvo_hip_params p;
  bpvo_hip_default_params(&p);                          // AlgorithmParameters() (bpvo/types.cc:31-66)
  p.numPyramidLevels = levels;
  p.descriptor = desc == "intensity" ? BPVO_DESC_INTENSITY : BPVO_DESC_BITPLANES;
  p.lossFunction = desc == "intensity" ? BPVO_LOSS_HUBER : BPVO_LOSS_TUKEY;
  p.verbosity = BPVO_VERB_SILENT;

  const size_t npix = (size_t) rows * cols;
  std::vector<uint8_t> images;
  std::vector<float> disparities;
  if(!readRaw(dir + "/images.u8", images, 2 * (size_t) n_pairs * npix) || !readRaw(dir + "/disparities.f32", disparities, 2 * (size_t) n_pairs * npix)) {
    std::fprintf(stderr, "cannot read %s/images.u8 / disparities.f32 (%d pairs of %dx%d)\n", dir.c_str(), n_pairs, cols, rows);
    return 1;
  }

  int lo = 0, hi = 0;
  bpvo_hip_shard_range(n_pairs, 0, n_gpus, &lo, &hi);   // the largest block
  bpvo_hip_node* node = nullptr;
  int rc = bpvo_hip_node_create(&node, n_gpus, nullptr, K, baseline, rows, cols, &p, hi - lo);
  if(rc) { std::fprintf(stderr, "bpvo_hip_node_create: %d %s\n", rc, bpvo_hip_node_last_error(nullptr)); return 1; }

  std::vector<float> poses((size_t) n_pairs * 16), records((size_t) n_pairs * 32);
  std::vector<bpvo_hip_stats> stats((size_t) n_pairs * levels);
  double best_ms = 0.0;
  for(int rep = 0; rep < 2; ++rep) {                    // the first run also pays for lazy allocations
    const auto t0 = std::chrono::steady_clock::now();
    rc = bpvo_hip_node_batch_run(node, n_pairs, images.data(), disparities.data(), poses.data(), records.data(), stats.data());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if(rc) { std::fprintf(stderr, "bpvo_hip_node_batch_run: %d %s\n", rc, bpvo_hip_node_last_error(node)); bpvo_hip_node_destroy(node); return 1; }
    if(rep == 0 || ms < best_ms) best_ms = ms;
  }
  long iterations = 0;
  for(size_t i = 0; i < stats.size(); ++i) iterations += stats[i].numIterations;
  std::printf("%d pairs %dx%d on %d GPU(s): %.2f ms per batch (host buffers), %.1f pairs/s, %ld GN iterations (numIterations summed)\n",
              n_pairs, cols, rows, bpvo_hip_node_num_devices(node), best_ms, 1e3 * n_pairs / best_ms, iterations);
  bpvo_hip_node_destroy(node);
  if(!prefix.empty() && (!writeRaw(prefix + "_poses.f32", poses) || !writeRaw(prefix + "_records.f32", records))) return 1;
  return 0;
}
