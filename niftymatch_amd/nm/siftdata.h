// siftdata.h -- keypoint / descriptor / match container (drop-in for NiftyMatch src/gpu/sift/siftdata.h:20-111).
#ifndef __SIFTPOINT_H__
#define __SIFTPOINT_H__

#include "device_vector.h"
#include "lazy_count.h"

#define SIFT_VECTOR_SIZE 128
#define MAX_DESCRIPTORS 2048

struct SiftData {
    nm::device_vector<float> _desc;          //!< capacity x 128
    nm::device_vector<int> _match_indexes;   //!< capacity, -1 = no match
    nm::device_vector<float> _x;
    nm::device_vector<float> _y;
    float *_x_ptr;
    float *_y_ptr;
    int *_match_indexes_ptr;
    //! Number of keypoints. An int in every use as a number; while compute_descriptors' count has not been read back it is
    //! PENDING on the device (lazy_count.h) and the first read waits for it. (`int _num_items` in the reference.)
    nm::lazy_int _num_items;
    int _capacity;
    //! Extension (lazy_count.h): the running item count on the device, two words used alternately; _items_cur is the one that
    //! holds the count while _num_items is pending.
    nm::device_vector<int> _items_dev;
    int _items_cur;
    //! Extension: scratch of compute_sift_matches(A = this, ...), kept between calls and only ever grown, so that a match
    //! call neither allocates nor synchronises (the reference allocates two device_vectors per call,
    //! sift/siftfunctions.cu:21,28).
    nm::device_vector<int> _match_workspace;

    SiftData() : _x_ptr(nullptr), _y_ptr(nullptr), _match_indexes_ptr(nullptr), _num_items(0), _capacity(0), _items_cur(0) {}
    SiftData(int capacity);      //!< throws std::runtime_error for capacity <= 0
    ~SiftData();
    //! Copies are deep (copy_from) and re-point _x_ptr / _y_ptr / _match_indexes_ptr at their own vectors; the match
    //! scratch is NOT copied (a copy allocates its own at its first compute_sift_matches).
    SiftData(const SiftData &in) : _x_ptr(nullptr), _y_ptr(nullptr), _match_indexes_ptr(nullptr), _num_items(0), _capacity(0), _items_cur(0) { copy_from(in); }
    SiftData &operator=(const SiftData &in) { if (this != &in) copy_from(in); return *this; }

    void copy_from(const SiftData &in);                   //!< deep copy of all vectors
    void initialize_data(int capacity = MAX_DESCRIPTORS);
    void clear_data();
};

#endif
