// stand-in (tests/stubs/README.md): dataclasses/I3Matrix.h is a boost::numeric::ublas::matrix<double> frame object;
// the glue uses (i, j), size1() and size2() only
#pragma once
#include <cstddef>
#include <vector>
#include <icetray/I3FrameObject.h>
class I3Matrix : public I3FrameObject {
public:
    I3Matrix() : rows_(0), cols_(0) {}
    I3Matrix(std::size_t size1, std::size_t size2) : rows_(size1), cols_(size2), data_(size1 * size2, 0.) {}
    std::size_t size1() const { return rows_; }
    std::size_t size2() const { return cols_; }
    double &operator()(std::size_t i, std::size_t j) { return data_[i * cols_ + j]; }
    const double &operator()(std::size_t i, std::size_t j) const { return data_[i * cols_ + j]; }
private:
    std::size_t rows_, cols_;
    std::vector<double> data_;
};
I3_POINTER_TYPEDEFS(I3Matrix);
