/*
 * bpvo_hip/config_file.hpp — the reference's `key = value` configuration files for the facade:
 *   bpvo::ConfigFile                      (reference: bpvo/config_file.h:64-127, bpvo/config_file.cc:33-71)
 *   bpvo::AlgorithmParametersFromFile     (reference: AlgorithmParameters(std::string), bpvo/types.cc:68-107)
 *   *FromString helpers                   (reference: bpvo/types.cc:120-200)
 * Keys are case-insensitive, lines starting with '#' or '%' are comments, white space is stripped, a line that is not
 * `key=value` throws.  NOTE the file defaults differ from the constructor defaults of AlgorithmParameters (e.g. Huber
 * vs Tukey, CD5 vs CD3, gradientTolerance 1e-6 vs 1e-8, sigmaPriorToCensusTransform 0.5 vs -1, minValidDisparity 1 vs
 * 0.001) exactly as in the reference.
 */
#ifndef BPVO_HIP_CONFIG_FILE_HPP
#define BPVO_HIP_CONFIG_FILE_HPP

#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>
#include <strings.h>

#include "vo.hpp"

namespace bpvo {

inline bool icompare(const std::string& a, const std::string& b)            // bpvo/utils.cc:44-47
{
  return a.size() == b.size() ? !strncasecmp(a.c_str(), b.c_str(), a.size()) : false;
}

struct CaseInsenstiveComparator {                                           // bpvo/utils.h (same spelling as the reference)
  bool operator()(const std::string& a, const std::string& b) const { return strcasecmp(a.c_str(), b.c_str()) < 0; }
};

class ConfigFile {
 public:
  ConfigFile() {}
  explicit ConfigFile(const std::string& filename)
  {
    std::ifstream ifs(filename);
    if(!ifs.is_open()) throw Error("could not open file '" + filename + "'");
    parse(ifs);
  }

  template <typename T> T get(const std::string& name) const
  {
    const auto it = _data.find(name);
    if(it == _data.end()) throw Error("no key " + name);
    T ret;
    std::istringstream ss(it->second);
    if((ss >> ret).bad()) throw Error("failed to convert '" + it->second + "'");
    return ret;
  }
  template <typename T> T get(const std::string& name, const T& default_val) const
  {
    try { return get<T>(name); } catch(const std::exception&) { return default_val; }
  }
  template <typename T> ConfigFile& set(const std::string& name, const T& value)
  {
    std::ostringstream ss; ss << value; _data[name] = ss.str(); return *this;
  }
  bool has(const std::string& name) const { return _data.find(name) != _data.end(); }

 private:
  void parse(std::ifstream& ifs)
  {
    std::string line;
    while(!ifs.eof()) {
      std::getline(ifs, line);
      if(line.empty()) continue;
      if(line.front() == '#' || line.front() == '%') continue;
      line.erase(std::remove_if(line.begin(), line.end(), [](char c) { return std::isspace((unsigned char) c); }), line.end());
      // splitstr(line, '=') (bpvo/utils.cc:96-105: std::getline tokens, so "=3" has an empty key and is accepted, "a=" and
      // a whitespace-only line are malformed) — pinned against the reference's reader in tests/test_reference_pins_cpu.py
      std::vector<std::string> tokens;
      {
        std::stringstream ss(line);
        std::string token;
        while(std::getline(ss, token, '=')) tokens.push_back(token);
      }
      if(tokens.size() != 2) throw Error("Malformed ConfigFile line " + line);
      _data[tokens[0]] = tokens[1];
    }
  }
  std::map<std::string, std::string, CaseInsenstiveComparator> _data;
};

template <> inline std::string ConfigFile::get<std::string>(const std::string& name) const
{
  const auto it = _data.find(name);
  if(it == _data.end()) throw Error("no key " + name);
  return it->second;
}

inline LossFunctionType LossFunctionTypeFromString(const std::string& s)
{
  if(icompare("Huber", s)) return kHuber;
  if(icompare("Tukey", s)) return kTukey;
  if(icompare("L2", s)) return kL2;
  throw Error("unknown LossFunctionType");
}
inline InterpolationType InterpolationTypeFromString(const std::string& s)
{
  if(icompare("Linear", s)) return kLinear;
  if(icompare("Cosine", s)) return kCosine;
  if(icompare("CubicHermite", s)) return kCubicHermite;
  if(icompare("Cubic", s)) return kCubic;
  throw Error("unknown InterpolationType");
}
inline int DescriptorTypeFromString(const std::string& s)     // numeric DescriptorType; Intensity, IntensityAndGradient, Laplacian and BitPlanes are on the device path
{
  if(icompare("Intensity", s)) return BPVO_DESC_INTENSITY;
  if(icompare("BitPlanes", s)) return BPVO_DESC_BITPLANES;
  if(icompare("Gradient", s) || icompare("IntensityAndGradient", s)) return 0x31;
  if(icompare("DescriptorFields", s)) return 0x32;
  if(icompare("Latch", s)) return 0x34;
  if(icompare("CentralDifference", s)) return 0x35;
  if(icompare("Laplacian", s)) return 0x36;
  if(icompare("DescriptorFields2", s) || icompare("DescriptorFields2ndOrder", s)) return 0x33;
  throw Error("unknown DescriptorType");
}
inline VerbosityType VerbosityTypeFromString(const std::string& s)
{
  if(icompare("Iteration", s)) return kIteration;
  if(icompare("Final", s)) return kFinal;
  if(icompare("Silent", s)) return kSilent;
  if(icompare("Debug", s)) return kDebug;
  throw Error("unknown VerbosityType");
}
inline GradientEstimationType GradientEstimationTypeFromString(const std::string& s)
{
  if(icompare("CD3", s)) return kCentralDifference_3;
  if(icompare("CD5", s)) return kCentralDifference_5;
  throw Error("unknown GradientEstimationType");
}

/* AlgorithmParameters(std::string filename) (reference: bpvo/types.cc:68-107), key spellings included */
inline AlgorithmParameters AlgorithmParametersFromFile(const std::string& filename)
{
  ConfigFile cf(filename);
  AlgorithmParameters p;
  p.numPyramidLevels = cf.get<int>("numPyramidLevels", -1);
  p.minImageDimensionForPyramid = cf.get<int>("minImageDimensionForPyramid", 40);
  p.sigmaPriorToCensusTransform = cf.get<float>("sigmaPriorToCensusTransform", 0.5f);
  p.sigmaBitPlanes = cf.get<float>("sigmaBitPlanes", 0.5f);
  p.dfSigma1 = cf.get<float>("dfSigma1", 0.75f);
  p.dfSigma2 = cf.get<float>("dfSigma2", 1.75f);
  p.latchNumBytes = cf.get<int>("latchNumBytes", 1);
  p.latchRotationInvariance = cf.get<int>("latchRotationInvariance", 0);
  p.latchHalfSsdSize = cf.get<int>("latchHalfSsdSize", 1);
  p.centralDifferenceRadius = cf.get<int>("centralDifferenceRadius", 3);
  p.centralDifferenceSigmaBefore = cf.get<float>("centralDifferenceSigmaBefore", 0.75f);
  p.centralDifferenceSigmaAfter = cf.get<float>("CenteralDifferenceSigmaAfter", 1.75f);
  p.laplacianKernelSize = cf.get<int>("laplacianKernelSize", 1);
  p.maxIterations = cf.get<int>("maxIterations", 50);
  p.parameterTolerance = cf.get<float>("parameterTolerance", 1e-7f);
  p.functionTolerance = cf.get<float>("functionTolerance", 1e-6f);
  p.gradientTolerance = cf.get<float>("gradientTolerance", 1e-6f);
  p.relaxTolerancesForCoarseLevels = cf.get<int>("relaxTolerancesForCoarseLevels", 1);
  p.gradientEstimation = GradientEstimationTypeFromString(cf.get<std::string>("GradientEstimation", "CD5"));
  p.interp = InterpolationTypeFromString(cf.get<std::string>("Interpolation", "Linear"));
  p.lossFunction = LossFunctionTypeFromString(cf.get<std::string>("lossFunction", "Huber"));
  p.descriptor = DescriptorTypeFromString(cf.get<std::string>("descriptor", "Intensity"));
  p.verbosity = VerbosityTypeFromString(cf.get<std::string>("Verbosity", "Iteration"));
  p.minTranslationMagToKeyFrame = cf.get<float>("minTranslationMagToKeyFrame", 0.1f);
  p.minRotationMagToKeyFrame = cf.get<float>("minRotationMagToKeyFrame", 2.5f);
  p.maxFractionOfGoodPointsToKeyFrame = cf.get<float>("maxFractionOfGoodPointsToKeyFrame", 0.6f);
  p.goodPointThreshold = cf.get<float>("goodPointThreshold", 0.75f);
  p.minNumPixelsForNonMaximaSuppression = cf.get<int>("minNumPixelsForNonMaximaSuppression", 320 * 240);
  p.nonMaxSuppRadius = cf.get<int>("nonMaxSuppRadius", 1);
  p.minNumPixelsToWork = cf.get<int>("minNumPixelsToWork", 256);
  p.minSaliency = cf.get<float>("minSaliency", 0.1f);
  p.minValidDisparity = cf.get<float>("minValidDisparity", 1.0f);
  p.maxValidDisparity = cf.get<float>("maxValidDisparity", 512.0f);
  p.maxTestLevel = cf.get<int>("maxTestLevel", 0);
  p.withNormalization = cf.get<int>("withNormalization", 1);
  return p;
}

/* ToString (reference: bpvo/types.cc:109-264) — the strings the apps print, spelling included ("CenteralDifference") */
inline std::string ToString(LossFunctionType t)
{
  switch(t) { case kHuber: return "Huber"; case kTukey: return "Tukey"; case kL2: return "L2"; }
  return "Unknown";
}
inline std::string ToString(VerbosityType v)
{
  switch(v) { case kIteration: return "Iteration"; case kFinal: return "Final"; case kSilent: return "Silent"; case kDebug: return "Debug"; }
  return "Unknown";
}
inline std::string ToString(PoseEstimationStatus s)
{
  switch(s) {
    case kParameterTolReached: return "ParameterTolReached";
    case kFunctionTolReached: return "FunctionTolReached";
    case kGradientTolReached: return "GradientTolReached";
    case kMaxIterations: return "MaxIterations";
    case kSolverError: return "SolverError";
  }
  return "Unknown";
}
inline std::string ToString(KeyFramingReason r)
{
  switch(r) {
    case kLargeTranslation: return "LargeTranslation";
    case kLargeRotation: return "LargeRotation";
    case kSmallFracOfGoodPoints: return "SmallFracOfGoodPoints";
    case kNoKeyFraming: return "NoKeyFraming";
    case kFirstFrame: return "FirstFrame";
  }
  return "Unknown";
}
inline std::string ToString(DescriptorType t)
{
  switch(t) {
    case kIntensity: return "Intensity";
    case kIntensityAndGradient: return "IntensityAndGradient";
    case kDescriptorFieldsFirstOrder: return "DescriptorFields";
    case kDescriptorFieldsSecondOrder: return "DescriptorFields2ndOrder";
    case kBitPlanes: return "BitPlanes";
    case kLatch: return "Latch";
    case kCentralDifference: return "CenteralDifference";
    case kLaplacian: return "Laplacian";
  }
  return "Unknown";
}
inline std::string ToString(GradientEstimationType t)
{
  switch(t) { case kCentralDifference_3: return "CentralDifference_3"; case kCentralDifference_5: return "CentralDifference_5"; }
  return "Unknown";
}
inline std::string ToString(InterpolationType t)
{
  switch(t) { case kLinear: return "Linear"; case kCosine: return "Cosine"; case kCubic: return "Cubic"; case kCubicHermite: return "CubicHermite"; }
  return "Unknown";
}

/* stream output (reference: bpvo/types.cc:267-303,312-320,349-364), same labels and order */
inline std::ostream& operator<<(std::ostream& os, const AlgorithmParameters& p)
{
  os << "numPyramidLevels = " << p.numPyramidLevels << "\n";
  os << "minImageDimensionForPyramid = " << p.minImageDimensionForPyramid << "\n";
  os << "sigmaPriorToCensusTransform = " << p.sigmaPriorToCensusTransform << "\n";
  os << "sigmaBitPlanes = " << p.sigmaBitPlanes << "\n";
  os << "dfSigma1 = " << p.dfSigma1 << "\n";
  os << "dfSigma2 = " << p.dfSigma2 << "\n";
  os << "latchNumBytes = " << p.latchNumBytes << "\n";
  os << "latchRotationInvariance = " << p.latchRotationInvariance << "\n";
  os << "latchHalfSsdSize = " << p.latchHalfSsdSize << "\n";
  os << "centralDifferenceRadius = " << p.centralDifferenceRadius << "\n";
  os << "centralDifferenceSigmaBefore = " << p.centralDifferenceSigmaBefore << "\n";
  os << "centralDifferenceSigmaAfter = " << p.centralDifferenceSigmaAfter << "\n";
  os << "laplacianKernelSize = " << p.laplacianKernelSize << "\n";
  os << "maxIterations = " << p.maxIterations << "\n";
  os << "parameterTolerance = " << p.parameterTolerance << "\n";
  os << "functionTolerance = " << p.functionTolerance << "\n";
  os << "gradientTolerance = " << p.gradientTolerance << "\n";
  os << "relaxTolerancesForCoarseLevel = " << p.relaxTolerancesForCoarseLevels << "\n";
  os << "gradienEstimation: " << ToString(static_cast<GradientEstimationType>(p.gradientEstimation)) << "\n";
  os << "InterpolationType: " << ToString(static_cast<InterpolationType>(p.interp)) << "\n";
  os << "lossFunction = " << ToString(static_cast<LossFunctionType>(p.lossFunction)) << "\n";
  os << "verbosity = " << ToString(static_cast<VerbosityType>(p.verbosity)) << "\n";
  os << "minTranslationMagToKeyFrame = " << p.minTranslationMagToKeyFrame << "\n";
  os << "minRotationMagToKeyFrame = " << p.minRotationMagToKeyFrame << "\n";
  os << "maxFractionOfGoodPointsToKeyFrame = " << p.maxFractionOfGoodPointsToKeyFrame << "\n";
  os << "goodPointThreshold = " << p.goodPointThreshold << "\n";
  os << "minNumPixelsForNonMaximaSuppression = " << p.minNumPixelsForNonMaximaSuppression << "\n";
  os << "minNumPixelsToWork = " << p.minNumPixelsToWork << "\n";
  os << "minSaliency = " << p.minSaliency << "\n";
  os << "minValidDisparity = " << p.minValidDisparity << "\n";
  os << "maxValidDisparity = " << p.maxValidDisparity << "\n";
  os << "withNormalization = " << p.withNormalization << "\n";
  os << "maxTestLevel = " << p.maxTestLevel;
  return os;
}
inline std::ostream& operator<<(std::ostream& os, const OptimizerStatistics& s)
{
  os << "numIterations: " << s.numIterations << "\n"
     << "finalError: " << s.finalError << "\n"
     << "firstOrderOptimality: " << s.firstOrderOptimality << "\n"
     << "status: " << ToString(s.status);
  return os;
}
inline std::ostream& operator<<(std::ostream& os, const ImageSize& s) { return os << "[" << s.rows << "," << s.cols << "]"; }
/* the pose prints as four rows of four values (the reference streams an Eigen::Matrix4f: same row layout, Eigen pads columns) */
inline std::ostream& operator<<(std::ostream& os, const Result& r)
{
  for(int i = 0; i < 4; ++i) os << r.pose[4 * i] << " " << r.pose[4 * i + 1] << " " << r.pose[4 * i + 2] << " " << r.pose[4 * i + 3] << "\n";
  os << "isKeyFrame: " << std::boolalpha << r.isKeyFrame << std::noboolalpha << "\n";
  if(!r.optimizerStatistics.empty()) os << r.optimizerStatistics.front();
  return os;
}

/* WriteTrajectoryKittiFormat (reference: apps/eval_kitti.cc:43-59): the 3x4 of every pose, row-major, "%lf" */
inline bool WriteTrajectoryKittiFormat(const std::string& filename, const Trajectory& trajectory)
{
  FILE* fp = std::fopen(filename.c_str(), "w");
  if(!fp) return false;
  for(size_t i = 0; i < trajectory.size(); ++i) {
    const Matrix44& T = trajectory[i];
    std::fprintf(fp, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf\n", (double) T[0], (double) T[1], (double) T[2], (double) T[3],
                 (double) T[4], (double) T[5], (double) T[6], (double) T[7], (double) T[8], (double) T[9], (double) T[10], (double) T[11]);
  }
  std::fclose(fp);
  return true;
}

}  // namespace bpvo
#endif
