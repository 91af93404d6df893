// vo_perf — the reference's timing harness (apps/vo_perf.cc:52-140) on top of bpvo_hip/vo.hpp.
//
//   vo_perf <dir> <rows> <cols> <fx> <fy> <cx> <cy> <baseline> <num_frames> [intensity|bitplanes] [output_prefix]
//
// <dir> holds raw frames image_%05d.u8 (rows*cols bytes) and disparity_%05d.f32 (rows*cols floats) — the synthetic
// stand-in for the reference's Dataset classes (utils/, out of scope).  Like the reference it times addFrame() per
// frame (std::chrono, microseconds instead of the reference's millisecond Timer), records the iteration count at
// maxTestLevel and optionally dumps <prefix>_poses.txt, <prefix>_iterations.txt, <prefix>_time.txt.
#include <bpvo_hip/vo.hpp>

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

int main(int argc, char** argv)
{
  if(argc < 10) {
    std::fprintf(stderr, "usage: %s dir rows cols fx fy cx cy baseline num_frames [intensity|bitplanes] [output_prefix]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
  const float fx = (float) std::atof(argv[4]), fy = (float) std::atof(argv[5]), cx = (float) std::atof(argv[6]), cy = (float) std::atof(argv[7]);
  const float baseline = (float) std::atof(argv[8]);
  const int max_frames = std::atoi(argv[9]);
  const std::string desc = argc > 10 ? argv[10] : "intensity";
  const std::string output_fn = argc > 11 ? argv[11] : "";

  bpvo::AlgorithmParameters params;                    // like apps/vo_example.cc:46-58
  params.numPyramidLevels = 3;
  params.maxIterations = 100;
  params.parameterTolerance = 1e-6f;
  params.functionTolerance = 1e-6f;
  params.verbosity = bpvo::kSilent;
  params.lossFunction = bpvo::kHuber;
  params.descriptor = desc == "bitplanes" ? bpvo::kBitPlanes : bpvo::kIntensity;
  params.minTranslationMagToKeyFrame = 0.1f;
  params.minRotationMagToKeyFrame = 2.5f;
  params.maxFractionOfGoodPointsToKeyFrame = 0.7f;
  params.goodPointThreshold = 0.8f;
  const int maxTestLevel = params.maxTestLevel;

  const bpvo::Matrix33 K = {{fx, 0.0f, cx, 0.0f, fy, cy, 0.0f, 0.0f, 1.0f}};
  try {
    bpvo::VisualOdometry vo(K, baseline, bpvo::ImageSize(rows, cols), params);
    std::vector<int> iterations;
    std::vector<double> time_ms;
    std::vector<bpvo::Matrix44> poses;
    double total_time = 0.0;
    std::vector<uint8_t> I;
    std::vector<float> D;
    char name[64];
    int f_i = 0;
    for(; f_i < max_frames; ++f_i) {
      std::snprintf(name, sizeof(name), "/image_%05d.u8", f_i);
      if(!readRaw(dir + name, I, (size_t) rows * cols)) { std::printf("no more data\n"); break; }
      std::snprintf(name, sizeof(name), "/disparity_%05d.f32", f_i);
      if(!readRaw(dir + name, D, (size_t) rows * cols)) break;

      const auto t0 = std::chrono::steady_clock::now();
      bpvo::Result result = vo.addFrame(I.data(), D.data());
      const double tt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      total_time += tt / 1000.0;

      const int num_iters = result.optimizerStatistics[maxTestLevel].numIterations;
      std::printf("Frame %05d %8.3f ms @ %7.2f Hz %03d iters keyframe %d reason 0x%x num_points %d\n", f_i, tt,
                  (f_i + 1) / total_time, num_iters, (int) result.isKeyFrame, (int) result.keyFramingReason, vo.numPointsAtLevel());
      poses.push_back(result.pose);
      time_ms.push_back(tt);
      iterations.push_back(num_iters);
    }
    // error behaviour of the reference: nullptr image/disparity throws bpvo::Error (bpvo/vo.cc:68-69)
    bool threw = false;
    try { vo.addFrame(nullptr, nullptr); } catch(const bpvo::Error&) { threw = true; }
    if(!threw) { std::fprintf(stderr, "addFrame(nullptr) did not throw\n"); return 1; }

    if(!output_fn.empty()) {
      std::ofstream p(output_fn + "_poses.txt"), it(output_fn + "_iterations.txt"), tm(output_fn + "_time.txt"), tr(output_fn + "_path.txt");
      for(size_t i = 0; i < poses.size(); ++i) {
        for(int k = 0; k < 16; ++k) p << poses[i][k] << " ";
        p << "\n";
        it << iterations[i] << "\n";
        tm << time_ms[i] << "\n";
      }
      const bpvo::Trajectory& traj = vo.trajectory();   // Trajectory::writeCameraPath equivalent: camera centres
      for(size_t i = 0; i < traj.size(); ++i) tr << traj[i][3] << " " << traj[i][7] << " " << traj[i][11] << "\n";
    }
    std::printf("done: %d frames, %.2f Hz\n", f_i, f_i / total_time);
  } catch(const bpvo::Error& e) {
    std::fprintf(stderr, "bpvo::Error: %s\n", e.what());
    return 1;
  }
  return 0;
}
