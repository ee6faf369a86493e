/*
 * lidarshooter_group.h -- one LiDAR frame over the GPUs of a node, from C or C++ (no Python, no PyTorch): the
 * multi-GPU part of SURVEY.md section 8(e) behind the C ABI of lidarshooter_hip.h.  One process per GPU; the
 * collective is RCCL (librccl.so.1, loaded at ls_group_create; xGMI between the GPUs of a node).  The environment variable
 * LS_GROUP_RCCL_LIBRARY, when set, names the library to load instead (an absolute path: a site's own RCCL build); a library
 * named there and not loadable is an error (LS_ERR_NO_DEVICE), never a silent fall-back.
 *
 * The reference has no multi-GPU path ("no NCCL / MPI / Gloo"); this is new surface.  Two ways
 * to spread a stream of frames over `world` GPUs, every GPU holding the whole scene:
 *
 *   LS_GROUP_SHARDED      every rank traces its azimuth sector of EVERY frame (ls_tracer_set_shard) and leaves its
 *                         hit records in a fixed-capacity slot  [n u32 | pad to 64 B | ls_hit x capacity];  ONE
 *                         ncclAllGather of the slots per frame is the all-gatherv of hit records (the count travels
 *                         in the slot header; the 32-byte points are a function of (ray, t) and are rebuilt by
 *                         ls_expand_gathered_hits on every rank).  Latency of one frame goes down with `world`.
 *   LS_GROUP_INTERLEAVED  rank g traces the WHOLE raster of the frames f with f mod world == g; nothing is exchanged
 *                         on the frame path.  Throughput goes up with `world`; a frame's latency stays that of one GPU.
 *                         (A 128 x 4096 frame over 1 M triangles takes ~20 us on one MI355X and is bound by launch
 *                         and memory latency, not by work: sharding it cannot win what a collective costs; interleaving can.)
 *
 * Frames rotate over three sets of buffers, so the gather + rebuild of frame f overlap the tracing of f+1 and f+2.
 *
 * SHARDED, since round 4 ("per-set" mode, the default): the group holds one communicator PER BUFFER SET (the one made
 * from the id and two duplicates of it: ncclCommSplit, or a second id broadcast over the first), so that everything of a
 * frame -- the three launches of the trace, ncclAllGather, the rebuild of the cloud -- is consecutive work on ONE stream,
 * the tracer's stream for that set, with no event and no cross-stream wait; and with LS_OPT_FRAME_GRAPH (switched on by
 * the group) that work is captured once per set into a HIP graph: a frame is one hipGraphLaunch, the k_project node's
 * arguments patched first when a pose changed.  The three sets' collectives run on three communicators and three
 * streams; every rank issues them in the same order.  LS_GROUP_FLAG_ONE_COMMUNICATOR keeps round 3's arrangement (one
 * communicator on a collective stream of its own, an event per hand-over): nine runtime calls per frame instead of one.
 */
#ifndef LIDARSHOOTER_GROUP_H
#define LIDARSHOOTER_GROUP_H

#include "lidarshooter_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ls_group ls_group;

#define LS_GROUP_SHARDED 0
#define LS_GROUP_INTERLEAVED 1
#define LS_GROUP_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define LS_GROUP_SLOT_HEADER 64

/* ---- slot arithmetic: plain functions, usable (and tested) without a GPU ---------------------------------------- */
/* contiguous azimuth columns of `rank`: the first H mod world ranks get one column more */
void ls_group_shard_columns(uint32_t H, uint32_t world, uint32_t rank, uint32_t *first_az, uint32_t *n_az);
/* records per slot = rays of the largest shard */
uint32_t ls_group_slot_capacity(uint32_t V, uint32_t H, uint32_t world);
uint64_t ls_group_slot_bytes(uint32_t capacity);
/* host twins of what the GPU does with a slot (CPU consumers, tests): fill one / unpack `world` gathered ones in rank
 * order (= ascending azimuth sector; ray indices inside the records are global).  decode returns the record count. */
void ls_group_write_slot(void *slot, uint32_t capacity, const ls_hit *hits, uint32_t n);
uint32_t ls_group_decode_gathered(const void *gathered, uint32_t world, uint32_t capacity, ls_hit *out_hits);

/* ---- the group ---------------------------------------------------------------------------------------------------- */
/* rank 0 makes the id (ncclGetUniqueId) and ships it to the other ranks by any means (file, pipe, MPI, environment) */
int ls_group_unique_id(uint8_t id[LS_GROUP_ID_BYTES]);
/* Collective over all ranks.  `tr` is this rank's tracer (created on its own GPU, geometries may be added before or
 * after); the group puts it on a stream of its own, sets its shard (SHARDED) and owns its output buffers from now on.
 * On failure the tracer is left, or put back, as ls_group_destroy leaves it: on its own stream, on the full turn, on
 * its own output buffers -- usable as a single-GPU tracer (the fallback when RCCL is missing: LS_ERR_NO_DEVICE). */
int ls_group_create(const uint8_t id[LS_GROUP_ID_BYTES], uint32_t world, uint32_t rank, int mode, ls_tracer *tr, ls_group **out);
/* the same with flags (equal on every rank):
 *   LS_GROUP_FLAG_ONE_COMMUNICATOR  round 3's arrangement: one communicator, a collective stream, events between the streams
 *   LS_GROUP_FLAG_NO_GRAPH          per-set mode without LS_OPT_FRAME_GRAPH: five plain enqueues per frame on one stream */
#define LS_GROUP_FLAG_ONE_COMMUNICATOR 1u
#define LS_GROUP_FLAG_NO_GRAPH 2u
/*   LS_GROUP_FLAG_SIZED_GATHER      (per-set mode) the all-gather moves the FRONT of every slot only -- the header and as many
 *                                   records as the set's previous frame, three frames ago, needed on its largest rank, plus a
 *                                   quarter, in steps of capacity / 16 -- instead of the whole fixed-capacity slot: at the
 *                                   headline half of a slot is padding.  The size comes out of the gathered headers, so it is
 *                                   the same number on every rank; the host waits for that frame's rebuild before it enqueues
 *                                   the set's next frame (at most three frames in flight either way), and a change of size
 *                                   captures the set's graph anew.  A frame whose hits outgrow the headroom within three frames
 *                                   is truncated on every rank alike: ls_group_download_cloud returns LS_ERR_OUT_OF_RANGE for it
 *                                   (ls_group_info: LS_GROUP_INFO_TRUNCATED_FRAMES), never a short cloud as if it were complete.
 *                                   Off by default: the fixed slots are latency-bound at 1 M triangles. */
#define LS_GROUP_FLAG_SIZED_GATHER 4u
/* The arrangement (per-set communicators, frames as graphs) is agreed on by the ranks at create: each rank says what it can do
 * -- its communicator duplicates exist, its tracer found three concurrent streams (a timing calibration) -- the answers are
 * gathered over the first communicator and every rank takes the AND (LS_GROUP_INFO_ARRANGEMENT_MINE / _COMMON): a rank that
 * fell back alone would issue its collectives on another communicator than its peers and the group would hang.  How the
 * duplicates are made (ncclCommSplit, or a second id broadcast over the first communicator) is agreed on the same way before
 * the first of them, and every rank attempts every duplicate whatever became of the one before: no rank ever skips a
 * collective its peers are in.  (Flag 0x100 is a test hook: lidarshooter_hip_debug.h.) */
int ls_group_create_opts(const uint8_t id[LS_GROUP_ID_BYTES], uint32_t world, uint32_t rank, int mode, uint32_t flags, ls_tracer *tr,
                         ls_group **out);
void ls_group_destroy(ls_group *g);
/* One frame, after the caller's updateGeometry / commitScene on the tracer: nothing in it waits for the device.
 * SHARDED: every rank calls it for every frame.  INTERLEAVED: every rank calls it for every frame too; it returns 1
 * without doing anything on the ranks that do not own the frame (0 where the frame was traced). */
int ls_group_trace(ls_group *g, uint32_t frame_index);
/* 1 if this rank holds frame_index's cloud (always, when SHARDED) */
int ls_group_owns_frame(const ls_group *g, uint32_t frame_index);
/* The whole frame's cloud on this rank: device pointers (points32, hits, count word), complete once the group's
 * collective stream has drained (ls_group_synchronize) -- buffers are reused three frames later. */
int ls_group_cloud(ls_group *g, uint32_t frame_index, ls_frame *out);
/* Whether that frame's cloud is complete -- to be asked after ls_group_synchronize (or any wait that covers the frame) by a
 * caller that reads ls_group_cloud's device pointers itself: LS_OK, or LS_ERR_OUT_OF_RANGE when LS_GROUP_FLAG_SIZED_GATHER
 * truncated it (ls_group_download_cloud checks this itself), or when the frame's buffers have been reused.  Without the flag
 * every frame is complete.  Returns LS_ERR_NOT_COMMITTED while the frame's rebuild has not finished. */
int ls_group_frame_status(ls_group *g, uint32_t frame_index);
/* The same cloud copied to host memory (waits for the frame): points32 takes 32 bytes per point, hits 16 (either may
 * be NULL); returns the number of points or a negative ls_status.  capacity in points. */
long ls_group_download_cloud(ls_group *g, uint32_t frame_index, void *points32, void *hits, uint32_t capacity);
int ls_group_synchronize(ls_group *g);
/* What RCCL itself says about the group (bench.py's "rccl" object): */
#define LS_GROUP_INFO_RCCL_VERSION 1   /* ncclGetVersion (e.g. 22707)                                          */
#define LS_GROUP_INFO_COMM_RANKS 2     /* ncclCommCount of the group's communicator: how many ranks RCCL sees   */
#define LS_GROUP_INFO_COMM_DEVICE 3    /* ncclCommCuDevice: the HIP device the communicator is bound to          */
#define LS_GROUP_INFO_COMMUNICATORS 4  /* 3 in per-set mode, 1 otherwise, 0 when INTERLEAVED                      */
#define LS_GROUP_INFO_PER_SET 5        /* 1: per-set mode (see the top of this header)                            */
#define LS_GROUP_INFO_FRAME_GRAPH 6    /* the tracer's LS_INFO_FRAME_GRAPH_STATE: 0 off, 1 frames are graph launches, 2 refused */
#define LS_GROUP_INFO_GATHER_CAPACITY 7   /* records of every slot that travel per frame now (the slot capacity unless SIZED_GATHER) */
#define LS_GROUP_INFO_TRUNCATED_FRAMES 8  /* SIZED_GATHER: frames whose hits outgrew the gather -- those counted when their set was reused plus
                                           * the finished ones among the (at most three) frames still held                            */
#define LS_GROUP_INFO_ARRANGEMENT_MINE 9    /* what this rank could do: bit 0 a communicator per set, bit 1 three concurrent streams  */
#define LS_GROUP_INFO_ARRANGEMENT_COMMON 10 /* the AND over all ranks: what the group runs                                            */
#define LS_GROUP_INFO_COLLECTIVES_ARE_A_SHIM 11 /* 1: LS_GROUP_RCCL_LIBRARY named tests/shim/librccl_shim.so (several ranks on ONE
                                                 * device through shared memory: semantics, not speed) -- a run on it must say so   */
long ls_group_info(ls_group *g, int what);
const char *ls_group_last_error(const ls_group *g);

#ifdef __cplusplus
}
#endif
#endif /* LIDARSHOOTER_GROUP_H */
