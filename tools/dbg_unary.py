import sys, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as ge
pt=ge.load_lab(); pt.set_device(0)  # pt_debug_* live in the lab library
print("sqrt fast vs literal:", pt.unary_compare(3,2,0,1<<32))
print("inv fast vs literal:", pt.unary_compare(1,0,0,1<<32))
# locate mismatches of inv by scanning exponent blocks
bad=[]
for e in range(27,227):
    n,ex=pt.unary_compare(1,0,e<<23,1<<23)
    if n: bad.append((e,n,hex(ex)))
print(len(bad), bad[:6], bad[-3:])
xs=np.array([int(b[2],16) for b in bad[:8]],dtype=np.uint32).view(np.float32)
for fn in (0,1,2,3):
    print(fn, [hex(v) for v in pt.unary_map(fn,xs).view(np.uint32)])
