#!/usr/bin/env python3
"""Dump tf.signal.linear_to_mel_weight_matrix for the shapes this repository tests, so that the mel weight matrix
can be pinned against real TensorFlow (it cannot be in the build container: TensorFlow is absent there).

Run this where TensorFlow is installed (the reference pins tensorflow-gpu==2.2.0, requirements.txt:1; any TF 2.x has
the same function), from the repository root:

    python scripts/dump_tf_mel.py            # writes tests/golden/tf_mel_<M>_<F>_<sr>.npz

then `python -m pytest tests/test_oracle.py -k tf_mel` compares the repository's recipe (oracle and the product's
host routine iris_mel_weight_matrix) with the dumped matrices - the test is reported as SKIPPED while the files are
absent.  Exact parity with a particular TF build needs no code change either way: pass TF's matrix to the plan
(`FrontendPlan(..., mel_matrix=W)` / `iris_plan_create(..., mel_host)`), see INTEGRATION.md section 4.

Call site in the reference: transforms.py:55-56 (defaults lower_edge_hertz=125.0, upper_edge_hertz=3800.0, float32)."""
import os
import sys

import numpy as np

SHAPES = [(80, 257, 16000), (64, 513, 16000), (128, 1025, 22050), (40, 129, 16000)]  # (num_mel_bins, num_spectrogram_bins, sample_rate)


def main():
    import tensorflow as tf
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for m, f, sr in SHAPES:
        w = tf.signal.linear_to_mel_weight_matrix(m, f, sr).numpy()  # the reference passes no edges / dtype
        assert w.shape == (f, m) and w.dtype == np.float32
        path = os.path.join(out_dir, f"tf_mel_{m}_{f}_{sr}.npz")
        np.savez_compressed(path, w=w, tf_version=np.array(tf.__version__), lower_edge_hertz=125.0, upper_edge_hertz=3800.0)
        print("wrote", path, "tf", tf.__version__, "nnz", int((w != 0).sum()))


if __name__ == "__main__":
    sys.exit(main())
