// Exercises include/bpvo_hip/io.hpp without a GPU: PLY writer, trajectory writers, KITTI odometry error metric.
//   io_test <out_dir>    writes cloud.ply, traj.txt, path.txt, gt/00.txt, est/00.txt, plot_{tl,rl,ts,rs}.txt and prints
//   "ERR first_frame r_err t_err len speed" per segment.
#include <bpvo_hip/config_file.hpp>
#include <bpvo_hip/io.hpp>

#include <cmath>
#include <cstdio>
#include <sys/stat.h>

static bpvo::Matrix44 relMotion(float yaw, float tx, float tz)
{
  bpvo::Matrix44 T;
  T.fill(0.0f);
  T[0] = std::cos(yaw); T[2] = std::sin(yaw); T[5] = 1.0f; T[8] = -std::sin(yaw); T[10] = std::cos(yaw); T[15] = 1.0f;
  T[3] = tx; T[11] = tz;
  return T;
}

int main(int argc, char** argv)
{
  if(argc < 2) return 2;
  const std::string dir = argv[1];
  mkdir((dir + "/gt").c_str(), 0755);
  mkdir((dir + "/est").c_str(), 0755);

  std::vector<bpvo::PointWithInfo> pts(5);
  for(int i = 0; i < 5; ++i) {
    pts[i].xyzw[0] = 0.5f * i; pts[i].xyzw[1] = -1.0f * i; pts[i].xyzw[2] = 2.0f + i; pts[i].xyzw[3] = 1.0f;
    pts[i].rgba[0] = pts[i].rgba[1] = pts[i].rgba[2] = (uint8_t) (10 * i); pts[i].rgba[3] = 255;
    pts[i].weight = 0.1f * i;
  }
  if(!bpvo::ToPlyFile(dir + "/cloud.ply", pts, "io_test")) return 3;

  // ~1.2 km drive at about 1 m per frame on a gentle arc; the estimate has a small yaw and scale bias
  bpvo::Trajectory gt, est;
  for(int i = 0; i < 1200; ++i) {
    gt.push_back(relMotion(-0.002f, 0.0f, -1.0f));      // push_back takes the frame-to-frame motion and inverts it
    est.push_back(relMotion(-0.00205f, 0.0f, -1.01f));
  }
  if(!bpvo::WriteTrajectory(dir + "/traj.txt", gt) || !bpvo::WriteCameraPath(dir + "/path.txt", gt)) return 4;
  if(!bpvo::WriteTrajectoryKittiFormat(dir + "/gt/00.txt", gt) || !bpvo::WriteTrajectoryKittiFormat(dir + "/est/00.txt", est)) return 5;

  const std::vector<bpvo::kitti::Pose> G = bpvo::kitti::LoadPoses(dir + "/gt/00.txt"), E = bpvo::kitti::LoadPoses(dir + "/est/00.txt");
  if(G.size() != 1200 || E.size() != 1200) return 6;
  const std::vector<bpvo::kitti::Errors> errs = bpvo::kitti::CalcSequenceErrors(G, E);
  for(size_t i = 0; i < errs.size(); ++i)
    std::printf("ERR %d %.9g %.9g %.9g %.9g\n", errs[i].first_frame, errs[i].r_err, errs[i].t_err, errs[i].len, errs[i].speed);
  bpvo::kitti::RunKittiEvaluation(dir + "/gt", dir + "/est", dir + "/plot", 0, 0);
  try {
    bpvo::kitti::LoadPoses(dir + "/missing.txt");
    return 7;
  } catch(const bpvo::Error&) {
  }
  return 0;
}
