import re
s = open('/root/repo/scratch/proto_ssn.py').read()
old = s[s.index("        dvec = -g.copy()  # active: gradient step"):s.index("        alpha = 1.0")]
new = '''        dvec = -g.copy()  # active: gradient step
        if F.any():
            idx = np.where(F)[0]
            dvec[idx] = chol_skip_solve(H[np.ix_(idx, idx)], -g[idx])
'''
s = s.replace(old, new)
s = s.replace("def project_ssn(", '''def chol_skip_solve(Hm, b, tau=1e-10):
    n = len(b); L = np.zeros((n, n)); skip = np.zeros(n, bool)
    for j in range(n):
        v = Hm[j, j] - L[j, :j] @ L[j, :j]
        if v <= tau * max(Hm[j, j], 1e-300):
            skip[j] = True; L[j, :] = 0; L[:, j] = 0; L[j, j] = 1.0
            continue
        L[j, j] = np.sqrt(v)
        for i in range(j + 1, n):
            L[i, j] = (Hm[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    bb = np.where(skip, 0.0, b)
    z = np.linalg.solve(L, bb)
    z[skip] = 0
    x = np.linalg.solve(L.T, z)
    x[skip] = 0
    return x


def project_ssn(''')
open('/root/repo/scratch/proto_ssn.py','w').write(s)
