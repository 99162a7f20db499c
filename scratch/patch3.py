s = open('/root/repo/scratch/proto_ssn.py').read()
old = s[s.index("        alpha = 1.0\n"):s.index("        if not ok:")]
new = '''        alpha = 1.0
        if MODE == 'ratio':
            dvec = np.where(act, 0.0, dvec)
            neg = nonneg & F & (dvec < 0)
            hit = np.zeros(p, bool)
            if neg.any():
                ratios = np.where(neg, theta / np.where(neg, -dvec, 1.0), np.inf)
                amax = ratios.min()
                if amax < 1.0:
                    alpha = amax; hit = neg & (ratios <= amax * (1 + 1e-12))
        ok = False
        for ls in range(40):
            th = theta + alpha * dvec
            th = np.where(nonneg, np.maximum(th, 0), th)
            if MODE == 'ratio':
                th = np.where(act, 0.0, th)
                if ls == 0: th = np.where(hit, 0.0, th)
            fn, resn = fval(th)
            if fn <= f + 1e-4 * g @ (th - theta):
                ok = True; break
            alpha *= 0.5
'''
s = s.replace(old, new)
s = s.replace("DELTA=1e-8\n", "DELTA=1e-10\nMODE='proj'\n")
s = s.replace("if len(sys.argv)>1: DELTA=float(sys.argv[1])", "if len(sys.argv)>1: DELTA=float(sys.argv[1])\n    if len(sys.argv)>2: MODE=sys.argv[2]")
open('/root/repo/scratch/proto_ssn.py','w').write(s)
