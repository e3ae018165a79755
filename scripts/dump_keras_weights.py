"""For a machine WITH TensorFlow and the reference checkout: convert a checkpoint of the reference model (Keras .h5,
`model.load_weights(NAME)`, sj_train.py:467-469 / eval.py:42-65) into the .npz that
`challenge_amd.sj_train.load_keras_weights` / `main(--pretrain True)` read.  Nothing of this repository is needed to run it.

    cd <reference checkout>
    python /path/to/dump_keras_weights.py <weights.h5> <out.npz> [-- <the reference's sj_train flags, e.g. --v 9 --n_mels 80>]

The arrays are stored as '<index>|<keras weight name>' in `model.weights` order (= layer order of define_keras_model), so the
loader needs neither h5py nor Keras' naming rules.  (`np.savez(out, *model.get_weights())` gives an equivalent file.)"""
import sys

import numpy as np


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        sys.exit(2)
    h5, out = sys.argv[1], sys.argv[2]
    flags = sys.argv[4:] if len(sys.argv) > 3 and sys.argv[3] == "--" else sys.argv[3:]
    sys.argv = [sys.argv[0]] + flags
    import sj_train as ref  # the reference's module (run from its checkout)
    config = ref.ARGS().get()
    model = ref.get_model(config)
    model.load_weights(h5)
    arrays = {f"{i:04d}|{w.name}": w.numpy() for i, w in enumerate(model.weights)}
    np.savez(out, **arrays)
    print(f"{len(arrays)} arrays, {sum(a.size for a in arrays.values())} values -> {out}")


if __name__ == "__main__":
    main()
