"""GPU (-m gpu): several contexts in ONE process, for the life of the process - what the integration is (INTEGRATION.md 2 registers HIPVCM,
HIPPathTracing and HIPBidirectional next to each other, each with its own etx_hip_context; the reference keeps all its integrators alive for the
life of the app, sources/raytracer/app.hxx:63-83, and runs / stops / updates one from the GUI thread while the others exist,
sources/etx/rt/integrators/integrator.cxx:76-114,174-184).

Round 5's GPU suite died with the interpreter's heap overwritten right after two contexts had been rendered, read back and destroyed
(GPUTEST_r05). These tests hold the ground that fix stands on:
  * three contexts (VCM, path tracer, bidirectional) alive together, updated round-robin from one host thread and from two, destroyed in every
    order: each film equals the film of the same integrator rendered alone;
  * after every teardown the process does the allocator-heavy things the suite died in (canary mappings in the address ranges that were just
    given back, a gigabyte allocated and freed in 4 KB pieces, imports of compiled modules) - and is still alive and unmodified afterwards.
"""
import gc
import itertools
import mmap
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPP = 32
KINDS = ("vcm", "pt", "bdpt")


def make(etx, golden_dir, kind, flavour="classic"):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_%s_128.etxscene" % flavour))
    snap.samples = SPP
    snap.noise_threshold = 0.0
    if kind == "vcm":
        integ = etx.HIPVCM(snap)
        integ.options().update({"vcm-blue_noise": False})
    elif kind == "pt":
        integ = etx.HIPPathTracing(snap)
        integ.options().update({"bn": False})
    else:
        integ = etx.HIPBidirectional(snap)
        integ.options().update({"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
    return integ


def films_of(etx, integ):
    cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
    stats = integ.status()
    assert stats.completed_iterations == SPP and stats.overflow_flags == 0 and stats.nonfinite_dropped == 0
    assert np.isfinite(cam).all() and np.isfinite(light).all()
    return cam[..., :3] + light[..., :3]


def drive(integrators):
    """update() round-robin until every integrator has stopped (Integrator::update never blocks: one iteration to a free lane, or nothing)"""
    from etx_tracer_amd.integrator import State
    for integ in integrators:
        integ.run()
    while any(i.state() != State.Stopped for i in integrators):
        for integ in integrators:
            if integ.state() != State.Stopped:
                integ.update()


def allocator_stress():
    """What a process does after a teardown - the part of the suite the round-5 library did not survive. Returns the number of canary bytes that
    changed (a stray write into memory the process has given back and mapped again)."""
    size = 256 * 1024  # the size of a 128 x 128 float4 film buffer and of a CPython arena
    pattern = b"\x5a" * size
    canaries = []
    for _ in range(64):
        m = mmap.mmap(-1, size)
        m.write(pattern)
        canaries.append(m)
    pieces = [bytearray(4096) for _ in range(1 << 18)]  # 1 GB in 4 KB pieces, freed again
    del pieces
    gc.collect()
    import scipy.ndimage  # noqa: F401  (compiled modules: dlopen + their static initialisers)
    import scipy.special  # noqa: F401
    changed = 0
    for m in canaries:
        if m[:] != pattern:
            changed += int((np.frombuffer(m[:], dtype=np.uint8) != 0x5A).sum())
        m.close()
    return changed


@pytest.fixture(scope="module")
def alone(etx, golden_dir):
    """every integrator rendered alone, one context at a time"""
    films = {}
    for kind in KINDS:
        integ = make(etx, golden_dir, kind)
        integ.render()
        films[kind] = films_of(etx, integ)
        integ.context.close()
    assert allocator_stress() == 0
    return films


def same_film(a, b):
    # the same iterations of the same scene: the films differ by the order of the float atomics only
    scale = float(np.abs(b).mean())
    return float(np.abs(a - b).mean()) < 2.0e-4 * scale and float(np.abs(a - b).max()) < 5.0e-2 * max(float(np.abs(b).max()), 1.0)


@pytest.mark.parametrize("order", list(itertools.permutations(range(3))))
def test_three_contexts_round_robin_from_one_thread(etx, golden_dir, alone, order):
    integrators = [make(etx, golden_dir, kind) for kind in KINDS]
    drive(integrators)
    films = [films_of(etx, integ) for integ in integrators]
    for index in order:  # destroyed in every order
        integrators[index].context.close()
    assert allocator_stress() == 0
    for kind, film in zip(KINDS, films):
        assert same_film(film, alone[kind]), kind


@pytest.mark.parametrize("order", list(itertools.permutations(range(3))))
def test_three_contexts_driven_from_two_threads(etx, golden_dir, alone, order):
    """The GUI thread drives one integrator while a second host thread drives the other two (two windows, or a host that renders a preview next to
    the final frame): contexts share nothing but the device."""
    integrators = [make(etx, golden_dir, kind) for kind in KINDS]
    errors = []

    def worker(group):
        try:
            drive(group)
        except Exception as error:  # noqa: BLE001 - reported by the test's thread
            errors.append(error)

    threads = [threading.Thread(target=worker, args=(integrators[:2],)), threading.Thread(target=worker, args=(integrators[2:],))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    films = [films_of(etx, integ) for integ in integrators]
    # torn down from two threads at once as well: the first two contexts on a second thread, the third here
    first = threading.Thread(target=lambda: [integrators[i].context.close() for i in order[:2]])
    first.start()
    integrators[order[2]].context.close()
    first.join()
    assert allocator_stress() == 0
    for kind, film in zip(KINDS, films):
        assert same_film(film, alone[kind]), kind


def test_render_read_destroy_cycles_leave_the_heap_alone(etx, golden_dir):
    """The shape of the failure itself: two contexts one after the other (the interleaved halves of the parity tests), films read back into numpy
    buffers, contexts destroyed, buffers freed - twenty times, with the allocator stress after every cycle."""
    for cycle in range(20):
        for first in (0, 1):
            snap = etx.SceneSnapshot(os.path.join(golden_dir, "cornell_classic_128.etxscene"))
            snap.samples = 256
            integ = etx.HIPBidirectional(snap, first_iteration=first, iteration_stride=2)
            integ.options().update({"bdpt-mode": etx.api.BDPT_MODE_FULL, "bdpt-blue_noise": False})
            integ.render()
            cam, light = integ.film(etx.api.LAYER_CAMERA), integ.film(etx.api.LAYER_LIGHT)
            assert np.isfinite(cam).all() and np.isfinite(light).all()
            integ.context.close()
            del cam, light, integ, snap
        assert allocator_stress() == 0, "cycle %d" % cycle
