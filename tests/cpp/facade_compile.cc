// Compile-only check of the C++ facade: every public member of the reference's API surface is instantiated.
#include <bpvo_hip/vo.hpp>

// The call pattern of the reference's apps/vo_perf.cc:52-85 (`VisualOdometry(dataset.get(), params)`, `vo.addFrame(frame)`)
// with stand-ins for the dataset classes of libbpvo_utils (out of scope: utils/dataset.h): the templated overloads of
// bpvo/vo.h:49-60,76-80 must accept them unchanged.
namespace vo_perf_like {
struct Mat {
  const void* p;
  template <typename T> T* ptr() const { return static_cast<T*>(p); }
};
struct Calibration { bpvo::Matrix33 K; float baseline; };
struct Frame {
  Mat img, dmap;
  const Mat& image() const { return img; }
  const Mat& disparity() const { return dmap; }
};
struct Dataset {
  Calibration calib;
  Calibration calibration() const { return calib; }
  bpvo::ImageSize imageSize() const { return bpvo::ImageSize(64, 64); }
  std::unique_ptr<Frame> getFrame(int) const { return std::unique_ptr<Frame>(new Frame{{nullptr}, {nullptr}}); }
};
int run(const Dataset* dataset, const bpvo::AlgorithmParameters& params)
{
  auto vo = bpvo::VisualOdometry(dataset, params);                        // apps/vo_perf.cc:56
  bpvo::VisualOdometry vo2(dataset->calibration(), dataset->imageSize());  // bpvo/vo.h:49-52
  std::unique_ptr<Frame> frame = dataset->getFrame(0);
  bpvo::Result result = vo.addFrame(frame);                               // apps/vo_perf.cc:85 (vo.addFrame(frame.get()) there)
  bpvo::Result result2 = vo2.addFrame(frame.get());
  return (int) result.optimizerStatistics.size() + (int) result2.isKeyFrame;
}
}  // namespace vo_perf_like

int facade_surface()
{
  bpvo::AlgorithmParameters p;
  p.numPyramidLevels = 2;
  p.descriptor = bpvo::kBitPlanes;
  p.lossFunction = bpvo::kTukey;
  const bpvo::Matrix33 K = {{100.f, 0.f, 32.f, 0.f, 100.f, 32.f, 0.f, 0.f, 1.f}};
  bpvo::VisualOdometry vo(K, 0.1f, bpvo::ImageSize(64, 64), p);
  bpvo::Result r = vo.addFrame(nullptr, nullptr);
  bpvo::Result rs = vo.addFrame(nullptr, nullptr, bpvo::StereoParameters(64));   // stereo front-end on the device
  (void) rs;
  (void) vo.numPointsAtLevel();
  (void) vo.pointsAtLevel(-1).size();
  (void) vo.trajectory().size();
  vo.setOption("lanes", 1); (void) vo.getOption("lanes");
  auto dev = std::make_shared<bpvo::detail::Device>(K, 0.1f, bpvo::ImageSize(64, 64), p, 2, 1);
  bpvo::VisualOdometryFrame ref(dev, 0), cur(dev, 1);
  ref.setData(nullptr, nullptr); ref.setTemplate(); (void) ref.hasTemplate(); (void) cur.empty(); cur.clear(); (void) ref.numLevels();
  (void) ref.levelSize(1).numel(); (void) ref.numChannels(); (void) ref.image(0).size(); (void) ref.descriptorChannel(0, 0).size();
  (void) ref.points(0).size(); (void) ref.pixels(0).size(); (void) ref.jacobians(0).size();
  bpvo::VisualOdometryPoseEstimator est(dev);
  bpvo::Matrix44 T0, T1;
  T0.fill(0.f);
  std::vector<bpvo::OptimizerStatistics> st = est.estimatePose(&ref, &cur, T0, T1);
  (void) est.getFractionOfGoodPoints(0.8f);
  (void) est.getWeights().size();
  return (int) st.size() + (r.pointCloud ? (int) r.pointCloud->size() : 0) + (int) r.keyFramingReason;
}
