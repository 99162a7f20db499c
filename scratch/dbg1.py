import sys; sys.path.insert(0,'/root/repo/scratch'); sys.path.insert(0,'/root/repo')
from proto_ssn import *
c, y, _ = synth.sp_batch(5, 5, 32, 0)
for i,(A,yy) in enumerate(zip(c,-y)):
    p,r,it = project_ssn(A,yy)
    if it>=20:
        print("instance",i); project_ssn(A,yy,max_it=30,verbose=True); break
