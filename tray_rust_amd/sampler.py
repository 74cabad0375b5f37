"""The Sampler types of src/sampler/ as the host side sees them: what a thread_work would construct per worker
(exec/multithreaded.rs:74 writes `sampler::LowDiscrepancy::new(queue.block_dim(), spp)`), reduced to the parameters that select the
device code -- the samples themselves are generated on the GPU (include/trayhip.h: tray_scene_set_sampler).

    LowDiscrepancy(dim, spp)             src/sampler/ld.rs:20-31
    Uniform(dim)                         src/sampler/uniform.rs:15-20
    Adaptive(dim, min_spp, max_spp)      src/sampler/adaptive.rs:34-52

`Hip(sampler=...)` takes a callable (block_dim, spp) -> one of these; the default is LowDiscrepancy."""
from ._lib import lib

LOW_DISCREPANCY, UNIFORM, ADAPTIVE = 0, 1, 2


def _next_power_of_two(v):
    return int(lib().tray_round_spp(int(v)))


class LowDiscrepancy:
    KIND = LOW_DISCREPANCY

    def __init__(self, dim, spp):
        self.dim = (int(dim[0]), int(dim[1]))
        self.spp = _next_power_of_two(spp)
        if self.spp != spp:      # ld.rs:22-25
            print(f"Warning: LowDiscrepancy sampler requires power of two samples per pixel, rounding up to {self.spp}")
        self.min_spp = self.max_spp_ = self.spp

    def max_spp(self):
        return self.spp

    def dimensions(self):
        return self.dim


class Uniform:
    KIND = UNIFORM

    def __init__(self, dim):
        self.dim = (int(dim[0]), int(dim[1]))
        self.spp = self.min_spp = self.max_spp_ = 1

    def max_spp(self):       # uniform.rs:48
        return 1

    def dimensions(self):
        return self.dim


class Adaptive:
    KIND = ADAPTIVE

    def __init__(self, dim, min_spp, max_spp):
        self.dim = (int(dim[0]), int(dim[1]))
        self.min_spp, self.max_spp_ = _next_power_of_two(min_spp), _next_power_of_two(max_spp)
        if self.min_spp != min_spp:      # adaptive.rs:37-41
            print(f"Warning: Adaptive sampler requires power of two samples per pixel, rounding min_spp up to {self.min_spp}")
        if self.max_spp_ != max_spp:     # adaptive.rs:42-46
            print(f"Warning: Adaptive sampler requires power of two samples per pixel, rounding max_spp up to {self.max_spp_}")
        if self.max_spp_ < self.min_spp:
            raise ValueError("Adaptive: max_spp < min_spp (adaptive.rs:48 would underflow)")
        self.step_size = int(lib().tray_adaptive_step(self.min_spp, self.max_spp_))     # adaptive.rs:48
        self.spp = self.min_spp

    def max_spp(self):       # adaptive.rs:122
        return self.max_spp_

    def dimensions(self):
        return self.dim
