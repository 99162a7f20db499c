s = open('/root/repo/scratch/proto_ssn.py').read()
s = s.replace("def chol_skip_solve(Hm, b, tau=1e-10):", "DELTA=1e-8\ndef chol_skip_solve(Hm, b, tau=1e-14):\n    Hm = Hm + DELTA*max(Hm.diagonal().max(),1e-300)*np.eye(len(b))")
s = s.replace('if __name__ == "__main__":', 'if __name__ == "__main__":\n    import sys\n    if len(sys.argv)>1: DELTA=float(sys.argv[1])')
open('/root/repo/scratch/proto_ssn.py','w').write(s)
