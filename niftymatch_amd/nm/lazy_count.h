// lazy_count.h -- host values that the DEVICE produces, resolved at the first host-visible read.
//
// The reference's per-octave client loop learns its keypoint counts on the host: thrust::copy_if returns an iterator
// (sift/pyramidata.cu:84-91: one implicit synchronisation per level), _orientations[level] is re-created with that size and
// compute_descriptors adds it to SiftData::_num_items (sift/siftfunctions.cu:165-178). Every octave therefore drains the
// device before the host may issue the next one -- ~470 us of a 1080p frame's 810 us on this path (round 4).
// Here the counts stay on the device: the kernels of compute_orientations / compute_descriptors read them there (every grid
// is sized for an upper bound and its surplus workgroups return at once), one lane also writes them into mapped pinned host
// words, and the host-side numbers -- _orientations[level].size(), SiftData::_num_items -- become PENDING: the first time the
// host looks at one, it waits for the stream that produces it and reads the word. A client that never looks (the common
// loop: frame after frame, then compute_sift_matches, which reads the sizes on the device too) never waits; a client that
// does look sees exactly what the reference's client sees. NM_EAGER_COUNTS=1 restores one synchronisation per octave.
#ifndef __NM_LAZY_COUNT_H__
#define __NM_LAZY_COUNT_H__

#include <hip/hip_runtime_api.h>

#include <memory>

namespace nm {

//! Four host words a kernel fills (mapped pinned memory): [0..2] the raw per-level counts of an octave, [3] a running item
//! count. resolve() waits for the producing stream once and latches them.
struct pending_counts {
    const volatile int *host;
    hipStream_t stream;
    bool resolved;
    int values[4];
    pending_counts(const volatile int *h, hipStream_t s) : host(h), stream(s), resolved(false), values{0, 0, 0, 0} {}
    void resolve();            // defined in pyramidata.cpp (error convention of the C++ layer)
};

//! An int whose value may still be on its way from the device. Converts to int (resolving), assigns from int (which makes the
//! host value authoritative again). Drop-in for `int SiftData::_num_items` wherever it is used as a number; the one thing it
//! cannot do is travel through C varargs (printf("%d", data._num_items) needs an explicit int(...)).
class lazy_int {
public:
    lazy_int(int v = 0) : _v(v) {}
    lazy_int(const lazy_int &o) : _v(int(o)) {}
    lazy_int &operator=(const lazy_int &o) { _v = int(o); _p.reset(); return *this; }
    lazy_int &operator=(int v) { _p.reset(); _v = v; return *this; }
    operator int() const
    {
        if (_p) {
            _p->resolve();
            _v = _p->values[3];
            _p.reset();
        }
        return _v;
    }
    lazy_int &operator+=(int d) { return *this = int(*this) + d; }
    lazy_int &operator-=(int d) { return *this = int(*this) - d; }
    lazy_int &operator++() { return *this += 1; }
    int operator++(int) { const int v = *this; *this = v + 1; return v; }
    //! true while the value has not been read back (the device-side word is then the authority)
    bool pending() const { return bool(_p); }
    void defer(std::shared_ptr<pending_counts> p) { _p = std::move(p); }
    //! the host value WITHOUT resolving (meaningful only while !pending())
    int host_value() const { return _v; }

private:
    mutable int _v;
    mutable std::shared_ptr<pending_counts> _p;
};

}  // namespace nm

#endif
