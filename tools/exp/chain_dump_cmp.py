import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
    d = np.abs(x - y).max() if x.shape else abs(float(x) - float(y))
    s = np.abs(y).max() if y.shape else abs(float(y))
    cos = float((x * y).sum() / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30)) if x.shape else 1.0
    if d > 0:
        print("%-44s max|d| %.3e  rel %.3e  cos %.6f" % (k, d, d / (s + 1e-30), cos))
