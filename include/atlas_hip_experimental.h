/*
 * atlas_hip_experimental.h — entry points libatlas_hip.so exports that are NOT part of the product interface (include/atlas_hip.h).
 *
 * What is here has not met the bar of the product header: parity-green on the hardware configuration it is meant for. Nothing in
 * `HipDistributedIndex`'s default configuration, `bench.py`'s default line or INTEGRATION.md's stub calls it. An entry point moves to
 * atlas_hip.h once it has; it may change or disappear without an ABI bump until then. Same conventions as atlas_hip.h (plain pointers
 * and sizes, caller-owned device memory, asynchronous on `stream`, 0 / hipError_t / ATLAS_E_* return values).
 */
#ifndef ATLAS_HIP_EXPERIMENTAL_H
#define ATLAS_HIP_EXPERIMENTAL_H

#include "atlas_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- peer exchange: a one-hop alternative to the all-gather + atlas_merge_packed above (src/index.py:134-151) ----------
 * Status: it has NEVER RUN ACROSS TWO DEVICES (no multi-GPU box was attached to the build); its logic is tested with two processes on
 * one GPU (tests/test_gpu_peer_exchange.py) and tests/test_rccl_multi.py runs it on W = 2 / 4 / 8 GPUs the day a node has them. Opt-in only:
 * HipDistributedIndex(exchange="peer"), bench.py --exchange peer. Every rank owns an exchange buffer that its peers map through a
 * 64-byte IPC handle (hipIpcMemHandle_t, carried as bytes by any channel, e.g. torch.distributed.all_gather_object):
 *   atlas_xchg_create   allocates and zeroes a buffer for W ranks x slot_entries packed candidates (>= B*k), returns its handle
 *   atlas_xchg_open     maps a PEER's buffer from its handle (the own buffer is used directly);  _close / _destroy undo them
 *   atlas_xchg_push     copies this rank's [n] packed candidates into slot `rank` of every buffer in peer_bufs[W] (host array of the
 *                       mapped pointers, own buffer at index rank) and then publishes `tag` there (system-scope release)
 *   atlas_xchg_merge    waits -- at most wait_ms -- until all W ranks have published `tag` in the own buffer, then writes the k best of the
 *                       W*k candidates of each of the B queries to out_packed (as atlas_merge_packed). If a peer is late, *status |= 1
 *                       (device int32, zeroed by the caller) and out_packed is left alone: repeat the exchange with the collective.
 * `tag` is the same nonzero number on every rank for one search and changes by one from search to search (slots are double-buffered
 * by its parity). All calls are asynchronous on `stream`. */
size_t atlas_xchg_bytes(int W, int64_t slot_entries);
int atlas_xchg_create(int W, int64_t slot_entries, void** buf, unsigned char* handle64);
int atlas_xchg_open(const unsigned char* handle64, void** peer_buf);
int atlas_xchg_close(void* peer_buf);
int atlas_xchg_destroy(void* buf);
int atlas_xchg_push(const uint64_t* packed, int64_t n, void* const* peer_bufs, int W, int rank, int64_t slot_entries, uint32_t tag, void* stream);
int atlas_xchg_merge(const void* own_buf, int W, int B, int k, int64_t slot_entries, uint32_t tag, int wait_ms, uint64_t* out_packed,
                     int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATLAS_HIP_EXPERIMENTAL_H */
