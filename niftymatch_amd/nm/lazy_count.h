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

#include <atomic>
#include <memory>
#include <mutex>

namespace nm {

//! Four host words a kernel fills (mapped pinned memory): [0..2] the raw per-level counts of an octave, [3] a running item
//! count. The ring pre-sets the words to -1 (counts are never negative) when it hands the slot out; resolve() waits ONCE until
//! the words this record's kernel writes (`need`, a bit per word) have left -1, and latches them. No stream handle and no event
//! is involved: the producing stream may be destroyed or its handle re-used at any time, and nothing is added to the stream
//! (an event recorded per launch cost the one-thread client loop 7 % on MI355X: 697 against 748 pairs/s). Thread-safe: any
//! number of host threads may resolve / read the same record concurrently.
struct pending_counts {
    volatile int *host;
    unsigned need;              //!< bit i: word i is written by this record's kernel
    int values[4];
    pending_counts(volatile int *h, unsigned need_words) : host(h), need(need_words), values{0, 0, 0, 0}, _resolved(false) {}
    void resolve();             //!< defined in pyramidata.cpp (error convention of the C++ layer)
    bool resolved() const { return _resolved.load(std::memory_order_acquire); }
    //! have the needed words arrived? (no waiting)
    bool arrived() const
    {
        for (int i = 0; i < 4; ++i)
            if (((need >> i) & 1u) && host[i] == -1) return false;
        return true;
    }

private:
    std::atomic<bool> _resolved;
    std::mutex _m;
};

//! An int whose value may still be on its way from the device. Converts to int (resolving), assigns from int (which makes the
//! host value authoritative again). Drop-in for `int SiftData::_num_items` wherever it is used as a number. The constructor from
//! int is EXPLICIT: with an implicit one, `cond ? data._num_items : 0` was ambiguous and mixed comparisons had two
//! viable conversions; now every mixed expression converts the lazy_int to int, as the reference's plain int would read. What a
//! class cannot do (INTEGRATION.md section 3.1 lists the patterns and their one-line fixes): bind to `int &` / `int *`, deduce as `int` in
//! std::min / std::max, travel through C varargs.
//! Reading is const and does not modify the object (a resolved record stays attached), so concurrent readers do not race;
//! writers need the same external ordering a plain int needs.
class lazy_int {
public:
    lazy_int() : _v(0) {}
    explicit lazy_int(int v) : _v(v) {}
    lazy_int(const lazy_int &o) : _v(int(o)) {}
    lazy_int &operator=(const lazy_int &o) { const int v = int(o); _p.reset(); _v = v; return *this; }
    lazy_int &operator=(int v) { _p.reset(); _v = v; return *this; }
    operator int() const
    {
        if (_p) { _p->resolve(); return _p->values[3]; }
        return _v;
    }
    lazy_int &operator+=(int d) { return *this = int(*this) + d; }
    lazy_int &operator-=(int d) { return *this = int(*this) - d; }
    lazy_int &operator++() { return *this += 1; }
    lazy_int &operator--() { return *this -= 1; }
    int operator++(int) { const int v = *this; *this = v + 1; return v; }
    int operator--(int) { const int v = *this; *this = v - 1; return v; }
    //! true while the value has not been read back (the device-side word is then the authority)
    bool pending() const { return _p && !_p->resolved(); }
    void defer(std::shared_ptr<pending_counts> p) { _p = std::move(p); }
    //! the host value WITHOUT waiting (meaningful only while !pending())
    int host_value() const { return (_p && _p->resolved()) ? _p->values[3] : _v; }

private:
    int _v;
    std::shared_ptr<pending_counts> _p;
};

}  // namespace nm

#endif
