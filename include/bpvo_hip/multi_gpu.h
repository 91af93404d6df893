/* bpvo_hip — one process driving all GPUs of a node: batches of independent frame pairs sharded over the devices, one
 * RCCL gather of the result records (BASELINE.json config 5; SURVEY.md §8b `bpvo_hip_gather_poses`, §8e).
 *
 * The reference has no multi-device code; the unit it parallelises over is the frame pair (a VisualOdometryPoseEstimator
 * touches only its two frames, bpvo/vo_pose_estimator.cc:63-93), so pairs are split into contiguous blocks, one block,
 * one bpvo_hip_ctx and one host thread per GPU, and nothing crosses GPUs on the data path.  The only exchange is ONE
 * ncclGather of the fixed 32-float records (pose 3x4, per-level iterations and statuses) over xGMI.
 *
 * Implemented by bpvo_amd/csrc/libbpvo_hip_mgpu.so (links libbpvo_hip.so and librccl.so).  It is a separate library so
 * that hosts which bring their own communicator — bench.py uses torch.distributed, whose RCCL is bundled with torch —
 * never load a second copy of RCCL.  Same conventions as c_api.h: int status (0 = ok), no exceptions across the ABI,
 * host pointers filled before return.
 */
#ifndef BPVO_HIP_MULTI_GPU_H
#define BPVO_HIP_MULTI_GPU_H

#include "c_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bpvo_hip_node bpvo_hip_node;

/* Contiguous block [lo, hi) of pair indices owned by `rank` of `world`: pair i -> rank i / ceil(n / world)
 * (the same rule as bpvo_amd/distributed.py:shard_range). */
void bpvo_hip_shard_range(int n_total, int rank, int world, int* lo, int* hi);

/* n_devices contexts (devices[r], or 0..n-1 when devices == NULL), each sized for max_pairs_per_device pairs, and an RCCL
 * communicator over them (ncclCommInitAll).  Parameters as bpvo_hip_create. */
int bpvo_hip_node_create(bpvo_hip_node** out, int n_devices, const int* devices, const float K[9], float baseline,
                         int rows, int cols, const bpvo_hip_params* p, int max_pairs_per_device);
void bpvo_hip_node_destroy(bpvo_hip_node* node);
const char* bpvo_hip_node_last_error(const bpvo_hip_node* node);   /* node may be NULL: last create error */
int bpvo_hip_node_num_devices(const bpvo_hip_node* node);
bpvo_hip_ctx* bpvo_hip_node_ctx(bpvo_hip_node* node, int rank);      /* e.g. for bpvo_hip_set_warp_formulation */

/* bpvo_hip_batch_run of n_pairs pairs (host buffers laid out as c_api.h says: images A0,B0,A1,B1,...), rank r running
 * its block on its own host thread; then bpvo_hip_gather_records to rank 0.
 *   poses   [n_pairs][16]   in pair order
 *   records [n_pairs][32]   the gathered records as they arrived at the root (NULL: not wanted)
 *   stats   [n_pairs][numLevels] (NULL: not wanted) */
int bpvo_hip_node_batch_run(bpvo_hip_node* node, int n_pairs, const uint8_t* images, const float* disparities,
                            float* poses, float* records, bpvo_hip_stats* stats);

/* The gather alone: the first n_local[r] records of the last batch of every context -> one ncclGather (counts padded to
 * the largest block) into the root's device buffer -> all_host [sum n_local][32] in rank order.  Every rank's send is
 * queued from this thread inside one ncclGroupStart / ncclGroupEnd. */
int bpvo_hip_gather_records(bpvo_hip_node* node, const int* n_local, int root, float* all_host);

#ifdef __cplusplus
}
#endif
#endif
