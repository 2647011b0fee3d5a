// hostcopy.hpp -- the host side of the drop-in (host-pointer) entry points tlpk_update / tlpk_solve (round 5).
//
// Tulip hands the backend ordinary pageable Vector{Float64}s (/root/reference/src/KKT/KKT.jl:83,100); they cross PCIe through
// pinned staging buffers owned by the handle.  Until round 4 ONE thread copied every vector into / out of the staging area with
// memcpy, strictly before / after the device work: 13 GB/s on config C4 (74 MB per Newton step -> 5.7 ms against 1.2 ms of link
// time), 19 GB/s on the north-star LP.  Now the vectors are cut into pieces of <= 512 KB and a small persistent pool of host
// threads works through them: a piece is copied into the staging area (non-temporal stores: the CPU never reads it again) and its
// hipMemcpyAsync is issued by the SAME thread right away, so the link is busy from the first piece on; on the way back every
// piece's device-to-host copy is followed by an event, and a thread copies the piece out of the staging area as soon as its event
// has fired while later pieces are still on the link.
//
// Why no hipHostRegister cache on the caller's arrays (round-4 review, item 1d): a registration pins PHYSICAL pages.  Julia frees a
// large Vector with munmap; a later allocation may get the same virtual address on different pages, and a cached registration would
// then make the DMA engine read the old pages -- silently stale data.  Nothing at this boundary tells the library that an address
// range has been unmapped, so caller memory is never registered behind the caller's back.
#pragma once
#include <cstddef>
#include <functional>

namespace tlpk {

// fn(i) for i in [0, n): the calling thread and the pool's workers draw indices from a shared counter; returns when all are done.
// TLPK_COPY_THREADS = number of worker threads (default 4, 0 = the caller does everything); the pool is created on first use and lives
// until the process ends.  Workers spin for ~100 us after a job before they sleep: the calls of a Newton step follow each other closely.
void host_parallel_for(int n, const std::function<void(int)> &fn);
int host_copy_threads();          // workers + 1

// dst <- src (bytes): non-temporal 16-byte stores when dst is 16-byte aligned (staging buffers are), memcpy otherwise
void copy_to_staging(void *dst, const void *src, size_t bytes);

}  // namespace tlpk
