// Parses a reference-style .cfg file with the facade's ConfigFile / AlgorithmParametersFromFile and prints every field
// (no GPU needed: only bpvo_hip_default_params is called).
#include <bpvo_hip/config_file.hpp>
#include <cstdio>
#include <string>
#include <iostream>
#include <sstream>

int main(int argc, char** argv)
{
  if(argc < 2) return 2;
  if(argc >= 5 && std::string(argv[2]) == "get") {    // raw lookup: config_file_test <file> get <key> <default>
    try {
      bpvo::ConfigFile cf{std::string(argv[1])};
      std::printf("%s\n", cf.get<std::string>(argv[3], std::string(argv[4])).c_str());
      return 0;
    } catch(const bpvo::Error& e) {
      std::printf("ERROR %s\n", e.what());
      return 1;
    }
  }
  try {
    const bpvo::AlgorithmParameters p = bpvo::AlgorithmParametersFromFile(argv[1]);
    std::printf("numPyramidLevels %d\nsigmaPriorToCensusTransform %g\nsigmaBitPlanes %g\nmaxIterations %d\nparameterTolerance %g\n"
                "functionTolerance %g\ngradientTolerance %g\ngradientEstimation %d\ninterp %d\nlossFunction %d\ndescriptor %d\n"
                "verbosity %d\nminTranslationMagToKeyFrame %g\nminRotationMagToKeyFrame %g\ngoodPointThreshold %g\nminSaliency %g\n"
                "minValidDisparity %g\nmaxTestLevel %d\nwithNormalization %d\ncentralDifferenceSigmaAfter %g\n",
                p.numPyramidLevels, p.sigmaPriorToCensusTransform, p.sigmaBitPlanes, p.maxIterations, p.parameterTolerance,
                p.functionTolerance, p.gradientTolerance, p.gradientEstimation, p.interp, p.lossFunction, p.descriptor, p.verbosity,
                p.minTranslationMagToKeyFrame, p.minRotationMagToKeyFrame, p.goodPointThreshold, p.minSaliency, p.minValidDisparity,
                p.maxTestLevel, p.withNormalization, p.centralDifferenceSigmaAfter);
    if(argc > 2) {   // printing helpers (bpvo/types.cc:109-364)
      std::ostringstream ss;
      ss << p << "\n--\n" << bpvo::OptimizerStatistics() << "\n--\n" << bpvo::ImageSize(3, 4) << "\n--\n";
      bpvo::Result r;
      r.pose.fill(0.0f); r.pose[0] = r.pose[5] = r.pose[10] = r.pose[15] = 1.0f;
      r.optimizerStatistics.push_back(bpvo::OptimizerStatistics());
      ss << r << "\n--\n" << bpvo::ToString(bpvo::kSmallFracOfGoodPoints) << " " << bpvo::ToString(bpvo::kCentralDifference) << " "
         << bpvo::ToString(bpvo::kBitPlanes) << " " << bpvo::ToString(bpvo::kFunctionTolReached) << " " << bpvo::ToString(bpvo::kCubicHermite);
      std::cout << "PRINT\n" << ss.str() << std::endl;
    }
  } catch(const bpvo::Error& e) {
    std::printf("ERROR %s\n", e.what());
    return 1;
  }
  return 0;
}
