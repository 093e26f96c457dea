"""examples/mnistsvm.m on the device engine: one-vs-rest linear SVM (hinge loss) on MNIST digits.

    python examples/mnistsvm.py [train-images.idx3-ubyte train-labels.idx1-ubyte [digit [count]]]

With the two idx files the images are cropped by 4 pixels, scaled to [0, 1] and flattened exactly as
mnistsvm.m:61-72 / 188-256 do (`synth.read_idx3_images`, `synth.read_idx1_labels`); without them (the image
files are not part of the reference tree) a synthetic matrix of the same shape and sparsity stands in.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

digit = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 6000
if len(sys.argv) > 2:
    D = ap.synth.read_idx3_images(sys.argv[1], count)
    labels = ap.synth.read_idx1_labels(sys.argv[2], count)
    ell = np.where(labels == digit, 1.0, -1.0)  # mnistsvm.m:136-142
    rng = np.random.default_rng(1)
    x0, z0, u0 = rng.random(D.shape[1]), rng.random(count), rng.random(count)
else:
    q = ap.synth.mnist_like_problem(seed=1, m=count, n=400, digit=digit)
    D, ell, x0, z0, u0 = q["D"], q["ell"], q["x0"], q["z0"], q["u0"]
res = ap.linearsvm(D, ell, 0.5, dict(rho=1.0, objevals=1, x0=x0, z0=z0, u0=u0))  # mnistsvm.m:42-43, 80
pred = np.sign(D @ res["xopt"])
print(f"digit {digit} vs rest on {count} samples: {res['steps']} iterations in {res['runtime'] * 1e3:.1f} ms "
      f"(setup + loop {res['solverruntime'] * 1e3:.1f} ms), objective {res['objopt']:.4f}, "
      f"training accuracy {np.mean(pred == ell):.4f}")
