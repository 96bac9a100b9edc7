"""Helper of tests/test_gpu_checkpoint.py::test_checkpoint_crosses_processes: renders the first `cut` iterations of a VCM render in a process of
its own and writes the checkpoint.   python tests/checkpoint_child.py <snapshot> <spp> <cut> <out>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import etx_tracer_amd as etx
    snapshot, spp, cut, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    snap = etx.SceneSnapshot(snapshot)
    snap.samples = spp
    integ = etx.HIPVCM(snap)
    integ.options()["vcm-blue_noise"] = False
    integ.run()
    while integ._rendered < cut:
        integ.update()
    integ.save_checkpoint(out)
    assert integ.status().completed_iterations == cut
    integ.context.close()


if __name__ == "__main__":
    main()
