// Parses a reference-style .cfg file with the facade's ConfigFile / AlgorithmParametersFromFile and prints every field
// (no GPU needed: only bpvo_hip_default_params is called).
#include <bpvo_hip/config_file.hpp>
#include <cstdio>

int main(int argc, char** argv)
{
  if(argc < 2) return 2;
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
  } catch(const bpvo::Error& e) {
    std::printf("ERROR %s\n", e.what());
    return 1;
  }
  return 0;
}
