// pyramidata.h -- per-frame scale-space buffers (drop-in for NiftyMatch src/gpu/sift/pyramidata.h:14-131).
// thrust::device_vector members are nm::device_vector (see device_vector.h).
#ifndef __PYRAMID_DATA_H__
#define __PYRAMID_DATA_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <memory>
#include <vector>

#include "device_vector.h"
#include "lazy_count.h"
#include "siftparams.h"

#define MAX_KERNEL_LENGTH 91

namespace nm {
//! Four pinned host ints for the asynchronous read-back of keypoint counts, allocated on first use. A COPY owns nothing
//! yet (it gets a buffer of its own on first use); a move transfers the buffer. This is what makes PyramidData freely
//! copyable and movable like the reference's (whose members are thrust vectors, sift/pyramidata.h:60-110): an implicit
//! member-wise copy of a raw pinned pointer would be freed twice and written to after the first free.
class pinned_counts {
public:
    pinned_counts() : _p(nullptr) {}
    pinned_counts(const pinned_counts &) : _p(nullptr) {}
    pinned_counts(pinned_counts &&o) noexcept : _p(o._p) { o._p = nullptr; }
    pinned_counts &operator=(const pinned_counts &) { return *this; }
    pinned_counts &operator=(pinned_counts &&o) noexcept
    {
        if (this != &o) { release(); _p = o._p; o._p = nullptr; }
        return *this;
    }
    ~pinned_counts() { release(); }
    int *get()
    {
        if (!_p) nm_check((int)hipHostMalloc(reinterpret_cast<void **>(&_p), 4 * sizeof(int), hipHostMallocDefault),
                          "pinned counter allocation");
        return _p;
    }
    bool allocated() const { return _p != nullptr; }

private:
    void release() { if (_p) (void)hipHostFree(_p); _p = nullptr; }
    int *_p;
};

//! A ring of NM_LAZY_SLOTS x 4 mapped pinned host words for lazy_count.h: a kernel writes a slot, the host reads it once the
//! words have left the -1 the ring put there. A slot is handed out again only once its last producer has WRITTEN it -- also
//! when that record was dropped unread: its kernel may still be about to write the words the next owner's kernel will write
//!. Same value semantics as pinned_counts: a copy starts with a ring of its own, a move takes the ring along.
class pinned_ring {
public:
    enum { SLOTS = 64 };
    pinned_ring() : _host(nullptr), _dev(nullptr), _next(0) { for (auto &n : _need) n = 0; }
    pinned_ring(const pinned_ring &) : _host(nullptr), _dev(nullptr), _next(0) { for (auto &n : _need) n = 0; }
    pinned_ring(pinned_ring &&o) noexcept : _host(o._host), _dev(o._dev), _next(o._next)
    {
        for (int i = 0; i < SLOTS; ++i) { _last[i] = std::move(o._last[i]); _need[i] = o._need[i]; }
        o._host = nullptr; o._dev = nullptr;
    }
    pinned_ring &operator=(const pinned_ring &) { return *this; }
    pinned_ring &operator=(pinned_ring &&o) noexcept
    {
        if (this != &o) {
            release();
            _host = o._host; _dev = o._dev; _next = o._next;
            for (int i = 0; i < SLOTS; ++i) { _last[i] = std::move(o._last[i]); _need[i] = o._need[i]; }
            o._host = nullptr; o._dev = nullptr;
        }
        return *this;
    }
    ~pinned_ring() { release(); }
    //! a fresh pending record for a kernel about to be launched on `stream` that writes the words `need` (bit per word);
    //! *dev_words = the slot's four words as the DEVICE addresses them
    std::shared_ptr<pending_counts> take(hipStream_t stream, unsigned need, int **dev_words);

private:
    void release();
    int *_host, *_dev;
    int _next;
    std::weak_ptr<pending_counts> _last[SLOTS];
    unsigned _need[SLOTS];             //!< the words the slot's last producer writes (0: never handed out)
};
}  // namespace nm

//! Upper bound of the keypoints of one DoG level of a w x h octave: strict maxima are pairwise non-adjacent, strict minima too.
inline size_t nm_keypoint_bound(int w, int h) { return 2 * (size_t)((w + 1) / 2) * (size_t)((h + 1) / 2); }

class PyramidData {
public:
    PyramidData() : _base_radius(0), _num_octaves(0), _num_dogs(0), _num_kernels(0), _lazy_octave(-1) { for (auto &d : _dirty) d = 0; }
    PyramidData(const SiftParams &params);
    // copy / move / destruction: member-wise (every member owns its memory and knows how to copy itself)

    void initialize(const SiftParams &params);
    void clear();
    //! Stable compaction of the valid entries among the first \c num_pixels of _key_pts[level] into
    //! _collated_kpts[level]; _orientations[level] is re-created with one (-1,-1) per surviving keypoint.
    void gpu_collate_keypoints_for_level(int level, int num_pixels);

public:
    nm::device_vector<float> _octave[20];          //!< Gaussian levels of the current octave
    nm::device_vector<float> _dog[19];             //!< differences of Gaussians
    nm::device_vector<float4> _key_pts[19];        //!< dense keypoint maps
    nm::device_vector<float2> _orientations[19];   //!< per collated keypoint
    nm::device_vector<float> _base_kernel;
    int _base_radius;
    nm::device_vector<float> _kernels[20];
    std::vector<int> _kernel_radii;
    nm::device_vector<float> _buffer;
    nm::device_vector<float2> _grad;
    nm::device_vector<float4> _collated_kpts[19];
    int _num_octaves;   //!< (sic) number of Gaussian levels per octave
    int _num_dogs;
    int _num_kernels;

    //! Extension used by compute_keypoints / compute_orientations: the three levels of an octave collated by one
    //! batched compaction and ONE device-to-host copy of the three counts (the reference's copy_if synchronises once
    //! per level). counts[l] = valid entries among the first \c num_pixels of _key_pts[l]; _collated_kpts[l] is filled.
    //! Waits for \c stream. _orientations is not touched.
    void gpu_collate_keypoints_for_octave(int num_pixels, int counts[3], hipStream_t stream = 0);
    //! compute_keypoints bookkeeping: _key_pts[l][i] may differ from -1 only for i < _dirty[l].
    size_t _dirty[19];

    //! Lazy counts (lazy_count.h): what compute_orientations left pending for compute_descriptors of the same octave -- the
    //! record its three _orientations sizes wait on, and the octave. Reset by any eager collation.
    std::shared_ptr<nm::pending_counts> _lazy_rec;
    int _lazy_octave;
    //! device words of the collation counts (3 ints, as nm_compact_keypoints3 leaves them)
    int *lazy_counts_dev() { return _count.data(); }
    //! the three-level collation without reading the counts back (they stay in lazy_counts_dev())
    void gpu_collate_keypoints_for_octave_dev(int num_pixels, hipStream_t stream);
    nm::pinned_ring _ring;

private:
    void generate_kernels(const SiftParams &params);
    void create_kernel_for_sigma(float sigma, nm::device_vector<float> &result, int &radius);
    nm::device_vector<int> _count;        // device-side counters + scratch of the compaction
    nm::device_vector<int> _compact_ws;
    nm::pinned_counts _host_counts;       // pinned, 4 ints
};

#endif
