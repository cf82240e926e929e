// siftparams.h -- scale-space constants of the SIFT path (drop-in for NiftyMatch src/gpu/sift/siftparams.h:14-99).
// Public field names and the float/double promotion of every formula follow the reference (siftparams.h:30-51),
// because the derived sigmas fix the Gaussian taps and therefore every downstream bit.
#ifndef __SIFT_PARAMS_H__
#define __SIFT_PARAMS_H__

#include <algorithm>
#include <cmath>
#include <vector>

#define MINIMUM_OCTAVE_SIZE 32

class SiftParams {
public:
    SiftParams() : _width(0), _height(0) {}

    SiftParams(int width, int height)
        : _width(width), _height(height), _num_dog_levels(3), _sigma_n(0.5f), _peak_threshold(0),
          _edge_threshold(10.f)
    {
        _level_min = -1;
        _level_max = _num_dog_levels + 1;
        const double shortest = std::min(width, height) * 2.0 / MINIMUM_OCTAVE_SIZE;
        _num_octaves = (int)std::floor(std::log(shortest) / std::log(2.0));
        if (_num_octaves <= 0) _num_octaves = 1;

        _sigma_k = std::pow(2.0f, 1.0f / _num_dog_levels);                                  // float pow
        _sigma_0 = 1.6f * _sigma_k;
        _sigma_d_0 = _sigma_0 * std::sqrt(1.0 - 1.0 / (_sigma_k * _sigma_k));               // double, narrowed
        const float sa = _sigma_0 * std::pow(_sigma_k, _level_min);                         // pow(float,int) -> double
        const float sb = _sigma_n;
        if (sa > sb) _base_smooth = std::sqrt(sa * sa - sb * sb);
        for (int i = _level_min + 1; i <= _level_max; ++i) _sigmas.push_back(_sigma_d_0 * std::pow(_sigma_k, i));
    }

    int _width;
    int _height;
    int _num_octaves;      // number of octaves of the pyramid
    int _num_dog_levels;   // S = 3
    int _level_max;        // iterate i <= _level_max - 2
    int _level_min;        // iterate from _level_min + 1
    float _sigma_d_0;
    float _sigma_k;
    float _sigma_0;
    float _sigma_n;
    float _base_smooth;    // smoothing that takes the input (sigma_n) to level _level_min
    std::vector<float> _sigmas;   // incremental blur from level i to i+1
    float _peak_threshold;
    float _edge_threshold;
};

#endif
