// tools/rand_probe.cpp -- which HIP runtime calls disturb the libc rand() stream?  (diagnostic)
#include "../include/nbody_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
static void probe(const char* what) {
    std::srand(1);
    // nothing between srand and rand here: baseline
    std::printf("%-28s first draw after srand(1): %d\n", what, std::rand());
}
int main() {
    std::printf("fresh process first draw: %d (expect 1804289383)\n", std::rand());
    std::srand(1);
    int n = 0;
    nb_device_count(&n);
    std::printf("after nb_device_count       : %d (expect 1804289383 if untouched)\n", std::rand());
    std::srand(1);
    nb_device_info_t info;
    nb_device_info(0, &info);
    std::printf("after nb_device_info        : %d\n", std::rand());
    std::srand(1);
    void* p = nullptr; void* q = nullptr; void* v = nullptr;
    nb_alloc(&p, 1 << 20); nb_alloc(&q, 1 << 20); nb_alloc(&v, 1 << 20);
    std::printf("after nb_alloc x3           : %d\n", std::rand());
    std::srand(1);
    std::vector<float> h(1 << 18, 1.0f);
    nb_h2d(p, h.data(), 1 << 20, nullptr); nb_h2d(v, h.data(), 1 << 20, nullptr);
    std::printf("after nb_h2d x2             : %d\n", std::rand());
    std::srand(1);
    nb_set_softening_sq_f32(0.01f);
    nb_integrate_f32((float*)q, (const float*)p, (float*)v, 0.016f, 1.0f, 1024, 256, NB_MODE_STRICT, nullptr);
    nb_device_synchronize();
    std::printf("after first kernel launch   : %d\n", std::rand());
    std::srand(1);
    nb_integrate_f32((float*)p, (const float*)q, (float*)v, 0.016f, 1.0f, 1024, 256, NB_MODE_FAST, nullptr);
    nb_d2h(h.data(), p, 1 << 20, nullptr);
    std::printf("after 2nd launch + d2h      : %d\n", std::rand());
    std::srand(1);
    nb_event_t e; nb_event_create(&e); nb_event_record(e, nullptr); nb_event_synchronize(e);
    std::printf("after event create/record   : %d\n", std::rand());
    return 0;
}
