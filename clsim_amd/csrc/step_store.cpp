#include "step_store.h"

#include <algorithm>

namespace clsimhip {

void StepStore::insert(const clsimhip_step &step)
{
    const size_t index = step.num_photons;
    if (index >= bins_.size()) bins_.resize(index + 1);             // StepStore.h:104-118
    ++pending_[step.identifier];
    bins_[index].push_back(step);
    ++size_;
}

uint32_t StepStore::count(uint32_t identifier) const
{
    const auto it = pending_.find(identifier);
    return it == pending_.end() ? 0u : it->second;
}

size_t StepStore::pop_bunch(size_t size, clsimhip_step *out)
{
    const size_t real = std::min(size, size_);
    size_t popped = 0;
    for (auto &queue : bins_) {
        if (popped >= real) break;
        while (!queue.empty() && popped < real) {
            const clsimhip_step &s = queue.front();
            const auto it = pending_.find(s.identifier);
            if (it != pending_.end() && --(it->second) == 0) pending_.erase(it);
            out[popped++] = s;
            queue.pop_front();
        }
    }
    size_ -= popped;
    return popped;
}

void StepStore::pop_bunch_filled(size_t size, clsimhip_step *out, const clsimhip_step &fill)
{
    const size_t real = pop_bunch(size, out);
    for (size_t i = real; i < size; ++i) out[i] = fill;
}

} // namespace clsimhip
