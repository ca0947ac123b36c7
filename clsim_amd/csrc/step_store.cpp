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

// the same as insert() step by step, run by run: steps of one light source with one photon count (a cascade's steps all but the last)
// enter their FIFO in one range insertion and are counted once
void StepStore::insert_many(const clsimhip_step *steps, size_t n)
{
    size_t i = 0;
    while (i < n) {
        const uint32_t photons = steps[i].num_photons, identifier = steps[i].identifier;
        size_t j = i + 1;
        while (j < n && steps[j].num_photons == photons && steps[j].identifier == identifier) ++j;
        if (photons >= bins_.size()) bins_.resize(static_cast<size_t>(photons) + 1);
        std::deque<clsimhip_step> &queue = bins_[photons];
        queue.insert(queue.end(), steps + i, steps + j);
        pending_[identifier] += static_cast<uint32_t>(j - i);
        size_ += j - i;
        i = j;
    }
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
        if (queue.empty()) continue;
        // the front of this FIFO, as many as still fit: copied out in one go, identifiers counted run by run
        const size_t take = std::min(queue.size(), real - popped);
        std::copy(queue.begin(), queue.begin() + static_cast<std::ptrdiff_t>(take), out + popped);
        queue.erase(queue.begin(), queue.begin() + static_cast<std::ptrdiff_t>(take));
        for (size_t i = popped; i < popped + take;) {
            const uint32_t identifier = out[i].identifier;
            size_t j = i + 1;
            while (j < popped + take && out[j].identifier == identifier) ++j;
            const auto it = pending_.find(identifier);
            if (it != pending_.end()) {
                const uint32_t gone = static_cast<uint32_t>(j - i);
                if (it->second <= gone) pending_.erase(it);
                else it->second -= gone;
            }
            i = j;
        }
        popped += take;
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
