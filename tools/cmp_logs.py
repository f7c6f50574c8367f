import numpy as np, sys
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    d = np.abs(a[k] - b[k]) / (np.abs(a[k]) + 1e-12)
    print("%-26s max rel diff sync vs ahead %.2e at step %d;  #(> 1e-4) %d of %d" % (k, d.max(), int(d.argmax()), int((d > 1e-4).sum()), len(d)))
