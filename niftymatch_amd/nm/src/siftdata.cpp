// siftdata.cpp -- SiftData (reference: src/gpu/sift/siftdata.cu:3-57).
#include "../siftdata.h"

#include <stdexcept>

SiftData::SiftData(int capacity)
    : _x_ptr(nullptr), _y_ptr(nullptr), _match_indexes_ptr(nullptr), _num_items(0), _capacity(0), _items_cur(0)
{
    if (capacity <= 0) throw std::runtime_error("Invalid initialization of SIFT data");
    initialize_data(capacity);
}

SiftData::~SiftData() { clear_data(); }

void SiftData::copy_from(const SiftData &in)
{
    _desc = in._desc;
    _x = in._x;
    _y = in._y;
    _match_indexes = in._match_indexes;
    _x_ptr = _x.data();
    _y_ptr = _y.data();
    _match_indexes_ptr = _match_indexes.data();
    _capacity = in._capacity;
    _num_items = in._num_items;                   // (reads a pending count back: the copy starts from a host value)
    _items_dev = in._items_dev.size() ? nm::device_vector<int>(2) : nm::device_vector<int>();    // (empty containers own nothing)
    _items_cur = 0;
}

void SiftData::initialize_data(int capacity)
{
    clear_data();
    _desc = nm::device_vector<float>((size_t)SIFT_VECTOR_SIZE * capacity, 0.f);
    _match_indexes = nm::device_vector<int>((size_t)capacity, -1);
    _x = nm::device_vector<float>((size_t)capacity);
    _y = nm::device_vector<float>((size_t)capacity);
    _x_ptr = _x.data();
    _y_ptr = _y.data();
    _match_indexes_ptr = _match_indexes.data();
    _capacity = capacity;
    _num_items = 0;
    _items_dev = nm::device_vector<int>(2);
    _items_cur = 0;
}

void SiftData::clear_data()
{
    _desc.clear();
    _match_indexes.clear();
    _x.clear();
    _y.clear();
    _x_ptr = _y_ptr = nullptr;
    _match_indexes_ptr = nullptr;
    _num_items = 0;
    _capacity = 0;
    _items_dev.clear();
    _items_cur = 0;
}
