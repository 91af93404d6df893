// libbpvo_hip_mgpu.so — include/bpvo_hip/multi_gpu.h: one process, one bpvo_hip_ctx + host thread per GPU, pairs sharded in
// contiguous blocks, ONE RCCL gather of the 32-float result records.  Host code only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpvo_hip/multi_gpu.h"

namespace {

constexpr int kRecordFloats = 32;
std::string g_create_err;

struct Rank {
  int device = 0;
  bpvo_hip_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  float* d_send = nullptr;      // [max_pairs][32] staging of this rank's records (padded block of the gather)
};

}  // namespace

struct bpvo_hip_node {
  std::vector<Rank> ranks;
  int max_pairs = 0;
  int rows = 0, cols = 0;
  float* d_recv = nullptr;      // on ranks[0].device: [world][max_pairs][32]
  std::vector<float> h_recv;
  std::string err;
};

static int node_fail(bpvo_hip_node* n, int code, const std::string& msg)
{
  if(n) n->err = msg;
  return code;
}

#define NODE_HIP(n, call)                                                                                   \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if(e_ != hipSuccess) return node_fail(n, BPVO_ERR_DEVICE, std::string(#call ": ") + hipGetErrorString(e_)); \
  } while(0)
#define NODE_NCCL(n, call)                                                                                   \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if(r_ != ncclSuccess) return node_fail(n, BPVO_ERR_DEVICE, std::string(#call ": ") + ncclGetErrorString(r_)); \
  } while(0)

extern "C" {

void bpvo_hip_shard_range(int n_total, int rank, int world, int* lo, int* hi)
{
  const int per = world > 0 ? (n_total + world - 1) / world : n_total;
  const int l = std::min(n_total, rank * per);
  if(lo) *lo = l;
  if(hi) *hi = std::min(n_total, l + per);
}

const char* bpvo_hip_node_last_error(const bpvo_hip_node* n) { return n ? n->err.c_str() : g_create_err.c_str(); }
int bpvo_hip_node_num_devices(const bpvo_hip_node* n) { return n ? (int) n->ranks.size() : 0; }
bpvo_hip_ctx* bpvo_hip_node_ctx(bpvo_hip_node* n, int rank)
{
  return (n && rank >= 0 && rank < (int) n->ranks.size()) ? n->ranks[rank].ctx : nullptr;
}

void bpvo_hip_node_destroy(bpvo_hip_node* n)
{
  if(!n) return;
  for(Rank& r : n->ranks) {
    (void) hipSetDevice(r.device);
    if(r.comm) (void) ncclCommDestroy(r.comm);
    if(r.stream) (void) hipStreamDestroy(r.stream);
    if(r.d_send) (void) hipFree(r.d_send);
    if(r.ctx) bpvo_hip_destroy(r.ctx);
  }
  if(n->d_recv && !n->ranks.empty()) {
    (void) hipSetDevice(n->ranks[0].device);
    (void) hipFree(n->d_recv);
  }
  delete n;
}

int bpvo_hip_node_create(bpvo_hip_node** out, int n_devices, const int* devices, const float K[9], float baseline, int rows,
                         int cols, const bpvo_hip_params* p, int max_pairs_per_device)
{
  if(!out) return BPVO_ERR_INVALID_ARG;
  *out = nullptr;
  int visible = 0;
  if(hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { g_create_err = "no HIP device"; return BPVO_ERR_NO_DEVICE; }
  if(n_devices <= 0 || n_devices > visible || max_pairs_per_device <= 0 || !K || !p) {
    g_create_err = "bpvo_hip_node_create: invalid argument (n_devices must be 1.." + std::to_string(visible) + ")";
    return BPVO_ERR_INVALID_ARG;
  }
  bpvo_hip_node* n = new bpvo_hip_node();
  n->max_pairs = max_pairs_per_device;
  n->rows = rows; n->cols = cols;
  n->ranks.resize(n_devices);
  std::vector<int> devs(n_devices);
  for(int r = 0; r < n_devices; ++r) devs[r] = n->ranks[r].device = devices ? devices[r] : r;
  auto bail = [&](int code, const std::string& msg) { g_create_err = msg; bpvo_hip_node_destroy(n); return code; };
  for(int r = 0; r < n_devices; ++r) {
    Rank& k = n->ranks[r];
    const int rc = bpvo_hip_create(&k.ctx, K, baseline, rows, cols, p, k.device, 2 * max_pairs_per_device, max_pairs_per_device);
    if(rc) return bail(rc, std::string("bpvo_hip_create on device ") + std::to_string(k.device) + ": " + bpvo_hip_last_error(nullptr));
    if(hipSetDevice(k.device) != hipSuccess || hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess ||
       hipMalloc((void**) &k.d_send, sizeof(float) * kRecordFloats * (size_t) max_pairs_per_device) != hipSuccess)
      return bail(BPVO_ERR_DEVICE, "stream / staging allocation failed on device " + std::to_string(k.device));
  }
  if(hipSetDevice(n->ranks[0].device) != hipSuccess ||
     hipMalloc((void**) &n->d_recv, sizeof(float) * kRecordFloats * (size_t) max_pairs_per_device * n_devices) != hipSuccess)
    return bail(BPVO_ERR_DEVICE, "gather buffer allocation failed");
  n->h_recv.resize((size_t) kRecordFloats * max_pairs_per_device * n_devices);
  std::vector<ncclComm_t> comms(n_devices);
  const ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, devs.data());
  if(nr != ncclSuccess) return bail(BPVO_ERR_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
  for(int r = 0; r < n_devices; ++r) n->ranks[r].comm = comms[r];
  *out = n;
  return BPVO_OK;
}

int bpvo_hip_gather_records(bpvo_hip_node* n, const int* n_local, int root, float* all_host)
{
  if(!n) return BPVO_ERR_INVALID_ARG;
  const int world = (int) n->ranks.size();
  if(!n_local || !all_host || root < 0 || root >= world) return node_fail(n, BPVO_ERR_INVALID_ARG, "bpvo_hip_gather_records: invalid argument");
  if(root != 0) return node_fail(n, BPVO_ERR_UNSUPPORTED, "the gather buffer lives on rank 0: root must be 0");
  int pad = 0;
  for(int r = 0; r < world; ++r) {
    if(n_local[r] < 0 || n_local[r] > n->max_pairs) return node_fail(n, BPVO_ERR_INVALID_ARG, "n_local out of range");
    pad = std::max(pad, n_local[r]);
  }
  if(pad == 0) return BPVO_OK;
  // stage every rank's records in its padded send block (device-to-device inside the rank, complete on return)
  for(int r = 0; r < world; ++r) {
    Rank& k = n->ranks[r];
    NODE_HIP(n, hipSetDevice(k.device));
    if(n_local[r] < pad)
      NODE_HIP(n, hipMemsetAsync(k.d_send, 0, sizeof(float) * kRecordFloats * (size_t) pad, k.stream));
    NODE_HIP(n, hipStreamSynchronize(k.stream));
    if(n_local[r] > 0) {
      const int rc = bpvo_hip_batch_copy_records_device(k.ctx, k.d_send, n_local[r]);
      if(rc) return node_fail(n, rc, std::string("batch_copy_records_device: ") + bpvo_hip_last_error(k.ctx));
    }
  }
  // the one collective: every rank contributes `pad` records, the root receives world * pad
  NODE_NCCL(n, ncclGroupStart());
  for(int r = 0; r < world; ++r) {
    Rank& k = n->ranks[r];
    const ncclResult_t g = ncclGather(k.d_send, n->d_recv, (size_t) kRecordFloats * pad, ncclFloat, root, k.comm, k.stream);
    if(g != ncclSuccess) {
      (void) ncclGroupEnd();
      return node_fail(n, BPVO_ERR_DEVICE, std::string("ncclGather: ") + ncclGetErrorString(g));
    }
  }
  NODE_NCCL(n, ncclGroupEnd());
  for(int r = 0; r < world; ++r) {
    NODE_HIP(n, hipSetDevice(n->ranks[r].device));
    NODE_HIP(n, hipStreamSynchronize(n->ranks[r].stream));
  }
  NODE_HIP(n, hipSetDevice(n->ranks[root].device));
  NODE_HIP(n, hipMemcpy(n->h_recv.data(), n->d_recv, sizeof(float) * kRecordFloats * (size_t) pad * world, hipMemcpyDeviceToHost));
  size_t off = 0;
  for(int r = 0; r < world; ++r) {
    std::memcpy(all_host + off * kRecordFloats, n->h_recv.data() + (size_t) r * pad * kRecordFloats, sizeof(float) * kRecordFloats * (size_t) n_local[r]);
    off += (size_t) n_local[r];
  }
  return BPVO_OK;
}

int bpvo_hip_node_batch_run(bpvo_hip_node* n, int n_pairs, const uint8_t* images, const float* disparities, float* poses,
                            float* records, bpvo_hip_stats* stats)
{
  if(!n) return BPVO_ERR_INVALID_ARG;
  const int world = (int) n->ranks.size();
  if(n_pairs <= 0 || !images || !disparities || !poses) return node_fail(n, BPVO_ERR_INVALID_ARG, "bpvo_hip_node_batch_run: invalid argument");
  std::vector<int> lo(world), hi(world), n_local(world), rcs(world, BPVO_OK);
  for(int r = 0; r < world; ++r) {
    bpvo_hip_shard_range(n_pairs, r, world, &lo[r], &hi[r]);
    n_local[r] = hi[r] - lo[r];
    if(n_local[r] > n->max_pairs) return node_fail(n, BPVO_ERR_INVALID_ARG, "more pairs per device than the node was created for");
  }
  const size_t npix = (size_t) n->rows * n->cols;
  const int L = bpvo_hip_num_levels(n->ranks[0].ctx);
  std::vector<float> local_poses((size_t) n_pairs * 16);
  std::vector<bpvo_hip_stats> local_stats((size_t) n_pairs * L);
  // one host thread per GPU; no data-path exchange between them
  std::vector<std::thread> threads;
  for(int r = 0; r < world; ++r) {
    if(n_local[r] == 0) continue;
    threads.emplace_back([&, r]() {
      rcs[r] = bpvo_hip_batch_run(n->ranks[r].ctx, n_local[r], images + 2 * (size_t) lo[r] * npix, disparities + 2 * (size_t) lo[r] * npix,
                                  0, local_poses.data() + 16 * (size_t) lo[r], local_stats.data() + (size_t) lo[r] * L);
    });
  }
  for(std::thread& t : threads) t.join();
  for(int r = 0; r < world; ++r)
    if(rcs[r]) return node_fail(n, rcs[r], std::string("rank ") + std::to_string(r) + ": " + bpvo_hip_last_error(n->ranks[r].ctx));
  if(stats) std::memcpy(stats, local_stats.data(), sizeof(bpvo_hip_stats) * local_stats.size());

  std::vector<float> rec((size_t) n_pairs * kRecordFloats);
  const int rc = bpvo_hip_gather_records(n, n_local.data(), 0, rec.data());
  if(rc) return rc;
  // the poses handed back are the gathered ones: record = pose 3x4 row-major first (c_api.h)
  for(int p = 0; p < n_pairs; ++p) {
    float* T = poses + 16 * (size_t) p;
    std::memcpy(T, rec.data() + (size_t) p * kRecordFloats, 12 * sizeof(float));
    T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
  }
  if(records) std::memcpy(records, rec.data(), sizeof(float) * rec.size());
  return BPVO_OK;
}

}  // extern "C"
