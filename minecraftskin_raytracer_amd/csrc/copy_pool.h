// copy_pool.h — a handful of worker threads for large host memcpys (host-only, no HIP): the rows of a host-buffer
// render land in a pinned ring and are copied into the caller's frame by several threads (api.cpp: download_staged).
#ifndef MCRT_COPY_POOL_H
#define MCRT_COPY_POOL_H

#include <stddef.h>

namespace mcrt {
// dst[0 .. bytes) = src[0 .. bytes), split over the pool's workers and the calling thread; returns when done.
// Safe to call from several threads (the jobs take turns).  MCRT_COPY_THREADS (default 8) sizes the pool.
void parallel_copy(void* dst, const void* src, size_t bytes);
}  // namespace mcrt

#endif
