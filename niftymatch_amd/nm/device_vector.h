// device_vector.h -- minimal owning device array used where the reference exposes thrust::device_vector members
// (sift/pyramidata.h:60-110, sift/siftdata.h:25-40). Documented deviation: no Thrust, no CUDA-compat shim.
// Interface subset: size/empty/clear/assign/resize/data/begin/end, construction from count (+ fill value) and from a
// host std::vector, deep copy, copy back to host. All operations use the current HIP device and the NULL stream and
// are synchronous, like thrust::device_vector's.
#ifndef __NM_DEVICE_VECTOR_H__
#define __NM_DEVICE_VECTOR_H__

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#include "exception.h"
#include "lazy_count.h"

extern "C" int nm_fill_u32(void *dst, size_t count, unsigned int pattern, void *stream);

namespace nm {

template <typename T>
class device_vector {
    static_assert(sizeof(T) % 4 == 0, "device_vector<T>: T must be a multiple of 4 bytes");

public:
    device_vector() : _p(nullptr), _n(0), _cap(0) {}
    explicit device_vector(size_t n) : _p(nullptr), _n(0), _cap(0) { allocate(n); zero(); }
    device_vector(size_t n, const T &value) : _p(nullptr), _n(0), _cap(0) { allocate(n); fill(value); }
    explicit device_vector(const std::vector<T> &host) : _p(nullptr), _n(0), _cap(0)
    {
        allocate(host.size());
        if (_n) nm_check((int)hipMemcpy(_p, host.data(), _n * sizeof(T), hipMemcpyHostToDevice), "device_vector H2D");
    }
    device_vector(const device_vector &o) : _p(nullptr), _n(0), _cap(0) { copy_from(o); }
    device_vector(device_vector &&o) noexcept : _p(o._p), _n(o._n), _cap(o._cap), _pend(std::move(o._pend)), _pend_idx(o._pend_idx)
    {
        o._p = nullptr; o._n = 0; o._cap = 0;
    }
    device_vector &operator=(const device_vector &o) { if (this != &o) copy_from(o); return *this; }
    device_vector &operator=(device_vector &&o) noexcept
    {
        if (this != &o) {
            release(); _p = o._p; _n = o._n; _cap = o._cap; _pend = std::move(o._pend); _pend_idx = o._pend_idx;
            o._p = nullptr; o._n = 0; o._cap = 0;
        }
        return *this;
    }
    ~device_vector() { release(); }

    //! (a deferred size is read through its record without modifying this object: concurrent readers do not race)
    size_t size() const { return _pend ? pending_size() : _n; }
    bool empty() const { return size() == 0; }
    T *data() { return _p; }
    const T *data() const { return _p; }
    T *begin() { return _p; }
    T *end() { return _p + size(); }
    const T *begin() const { return _p; }
    const T *end() const { return _p + size(); }
    //! Deferred size (lazy_count.h): size() will be word `index` of `p` once the device has produced it -- or 0 if one of the
    //! words before it is 0: an empty level ends its octave (sift/siftfunctions.cu:145). The allocation must already hold the
    //! largest possible count (reserve_uninitialized).
    void defer_size(std::shared_ptr<pending_counts> p, int index) { _pend = std::move(p); _pend_idx = index; }
    bool size_pending() const { return _pend && !_pend->resolved(); }
    //! the pending record (or null): lets the owner check that a deferred size is still the one it installed
    const std::shared_ptr<pending_counts> &pending_record() const { return _pend; }
    //! capacity >= n, size and contents untouched (no-op when already large enough; otherwise the contents are lost)
    void reserve_uninitialized(size_t n)
    {
        if (n > _cap) {
            const size_t keep = size();
            release();
            allocate(n);
            _n = keep < n ? keep : n;
        }
    }
    void clear() { release(); }
    void assign(size_t n, const T &value) { release(); allocate(n); fill(value); }
    void resize(size_t n) { if (n != size()) { release(); allocate(n); zero(); } }
    size_t capacity() const { return _cap; }
    //! size() becomes n WITHOUT touching the device: the allocation is kept while it is large enough (grow-only, with
    //! headroom), the contents are unspecified. For buffers a kernel fills completely right afterwards -- no
    //! hipMalloc / hipFree / fill on the per-octave path (the reference re-creates _orientations[level] per level,
    //! sift/pyramidata.cu:90).
    void resize_uninitialized(size_t n)
    {
        _pend.reset();
        if (n > _cap) {
            release();
            allocate(n + n / 2 + 64);
        }
        _n = n;
    }

    void fill(const T &value)
    {
        settle();
        if (!_n) return;
        unsigned int w[sizeof(T) / 4];
        std::memcpy(w, &value, sizeof(T));
        bool uniform = true;
        for (size_t i = 1; i < sizeof(T) / 4; ++i) uniform = uniform && (w[i] == w[0]);
        if (uniform) {
            nm_check(nm_fill_u32(_p, _n * (sizeof(T) / 4), w[0], nullptr), "device_vector fill");
            nm_check((int)hipStreamSynchronize(nullptr), "device_vector fill sync");
        } else {
            std::vector<T> h(_n, value);
            nm_check((int)hipMemcpy(_p, h.data(), _n * sizeof(T), hipMemcpyHostToDevice), "device_vector fill H2D");
        }
    }
    std::vector<T> to_host() const
    {
        const size_t n = size();
        std::vector<T> h(n);
        if (n) nm_check((int)hipMemcpy(h.data(), _p, n * sizeof(T), hipMemcpyDeviceToHost), "device_vector D2H");
        return h;
    }

private:
    void allocate(size_t n)
    {
        _n = n;
        _cap = n;
        _p = nullptr;
        if (n) nm_check((int)hipMalloc(reinterpret_cast<void **>(&_p), n * sizeof(T)), "device_vector alloc");
    }
    void zero()
    {
        if (_n) nm_check((int)hipMemset(_p, 0, _n * sizeof(T)), "device_vector zero");
    }
    void release()
    {
        if (_p) (void)hipFree(_p);
        _p = nullptr;
        _n = 0;
        _cap = 0;
        _pend.reset();
    }
    //! a deferred size as a number: waits for the producing launch (once per pending record); does not modify the vector
    size_t pending_size() const
    {
        _pend->resolve();
        bool live = true;
        for (int k = 0; k < _pend_idx; ++k) live = live && _pend->values[k] != 0;
        // an empty level ends its octave (sift/siftfunctions.cu:145): the levels behind it count as empty (SURVEY Q9 -- the
        // reference leaves their previous sizes in place, which nothing reads: compute_descriptors stops at the empty level too)
        const size_t n = (live && _pend->values[_pend_idx] > 0) ? (size_t)_pend->values[_pend_idx] : 0;
        return n < _cap ? n : _cap;
    }
    //! latch a deferred size (non-const paths that go on to change the vector)
    void settle()
    {
        if (!_pend) return;
        _n = pending_size();
        _pend.reset();
    }
    void copy_from(const device_vector &o)
    {
        const size_t n = o.size();
        release();
        allocate(n);
        if (_n) nm_check((int)hipMemcpy(_p, o._p, _n * sizeof(T), hipMemcpyDeviceToDevice), "device_vector D2D");
    }
    T *_p;
    size_t _n;
    size_t _cap;
    std::shared_ptr<pending_counts> _pend;
    int _pend_idx = 0;
};

}  // namespace nm

#endif
