// Host side of the step producers: the container that sorts steps by photon count and the bunching rule of the
// reference's feeder thread.  References: public/clsim/I3CLSimStepStore.h:44-320 (I3CLSimTemplateStore /
// I3CLSimStepStore), private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx:209-273 (flushStepStore, emitStep).
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <map>
#include <vector>

#include "../../include/clsimhip.h"

namespace clsimhip {

class StepStore {
public:
    explicit StepStore(size_t initial_bins) : bins_(initial_bins) {}

    // insert_copy(step.GetNumPhotons(), step): one FIFO per photon count, identifiers counted (StepStore.h:266-283)
    void insert(const clsimhip_step &step);
    void insert_many(const clsimhip_step *steps, size_t n);        // insert() for each of them, in order
    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    // steps of this identifier still in the store (StepStore.h:308-312)
    uint32_t count(uint32_t identifier) const;
    // pop_bunch_to_vector(size, vect): up to `size` steps in ascending photon count, first in first out within a
    // count (StepStore.h:163-198, 286-296); returns how many were written
    size_t pop_bunch(size_t size, clsimhip_step *out);
    // pop_bunch_to_vector(size, vect, temp): the remainder is filled with copies of `fill` (StepStore.h:209-222)
    void pop_bunch_filled(size_t size, clsimhip_step *out, const clsimhip_step &fill);
    // numStepsWithDummyFill (Async.cxx:256): what the last bunch before a barrier is padded to -- one whole granule
    // more even when the store already holds a multiple of the granularity
    size_t size_with_dummy_fill(size_t granularity) const
    {
        return granularity > 1 ? ((size_ / granularity) + 1) * granularity : size_;
    }

private:
    std::vector<std::deque<clsimhip_step>> bins_;
    std::map<uint32_t, uint32_t> pending_;
    size_t size_ = 0;
};

} // namespace clsimhip
