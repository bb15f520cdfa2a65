import numpy as np, sys
sys.path.insert(0,'.')
from oracle import oracle as O
import ctypes as C
M1=np.uint64(4294967087); M2=np.uint64(4294944443)
MASK=np.uint64(0xffffffff); S32=np.uint64(32)
def fold(x,c,m,n):
    c=np.uint64(c)
    for _ in range(n):
        x=(x>>S32)*c+(x&MASK)
    return np.where(x>=m,x-m,x)
def f1(x): return fold(x,209,M1,2)
def f2(x): return fold(x,22853,M2,3)
def matvec(A,v,f):
    r=[]
    for i in range(3):
        acc=f(np.uint64(A[3*i])*v[0])+f(np.uint64(A[3*i+1])*v[1])+f(np.uint64(A[3*i+2])*v[2])
        r.append(f(acc))
    return r
def matmul(A,B,m):
    A=[int(x) for x in A]; B=[int(x) for x in B]
    return [sum(A[3*i+k]*B[3*k+j] for k in range(3))%m for i in range(3) for j in range(3)]
def matpow(A,e,m):
    R=[1,0,0,0,1,0,0,0,1]
    while e:
        if e&1: R=matmul(R,A,m)
        A=matmul(A,A,m); e>>=1
    return R
m1=int(M1); m2=int(M2)
A1=[0,1,0,0,0,1,m1-810728,1403580,0]; A2=[0,1,0,0,0,1,m2-1370589,0,527612]
K,T=2048,100
L=4; Cn=25
# init states via oracle-equivalent exact python
seed=1234
x1=(seed&0xffffffff)^0x55555555; x2=((seed>>32)^0xAAAAAAAA)&0xffffffff
base1=[x1*12345%m1,x2*12345%m1,x1*12345%m1]; base2=[x2*12345%m2,x1*12345%m2,x2*12345%m2]
sub1=matpow(A1,2**76,m1); sub2=matpow(A2,2**76,m2)
# emulate init kernel: bits of k
k=np.arange(K,dtype=np.uint64)
s1=[np.full(K,b,dtype=np.uint64) for b in base1]; s2=[np.full(K,b,dtype=np.uint64) for b in base2]
P1=sub1; P2=sub2
for b in range(11):
    n1=matvec(P1,s1,f1); n2=matvec(P2,s2,f2)
    sel=((k>>np.uint64(b))&np.uint64(1))==1
    s1=[np.where(sel,a,o) for a,o in zip(n1,s1)]; s2=[np.where(sel,a,o) for a,o in zip(n2,s2)]
    P1=matmul(P1,P1,m1); P2=matmul(P2,P2,m2)
print('init unreduced?', any((np.any(a>=M1) for a in s1)), any((np.any(a>=M2) for a in s2)))
# check init vs exact
st=O.MrgState(); Lb=O.lib(); Lb.orc_mrg_seed(C.byref(st),1234)
ok=True
for kk in (0,1,5,336,2047):
    st=O.MrgState(); Lb.orc_mrg_seed(C.byref(st),1234); Lb.orc_mrg_skip_subsequences(C.byref(st),kk)
    if [int(s1[i][kk]) for i in range(3)]!=list(st.s1) or [int(s2[i][kk]) for i in range(3)]!=list(st.s2): ok=False; print('init mismatch',kk)
print('init ok',ok)
# jumps
bad=0
for c in range(1,Cn):
    J1=matpow(A1,2*L*c,m1); J2=matpow(A2,2*L*c,m2)
    n1=matvec(J1,s1,f1); n2=matvec(J2,s2,f2)
    for kk in range(K):
        pass
    # exact
    e1=[np.array([sum(J1[3*i+j]*int(s1[j][kk]) for j in range(3))%m1 for kk in range(K)],dtype=np.uint64) for i in range(3)]
    e2=[np.array([sum(J2[3*i+j]*int(s2[j][kk]) for j in range(3))%m2 for kk in range(K)],dtype=np.uint64) for i in range(3)]
    b=sum(int(np.sum(a!=b)) for a,b in zip(n1,e1))+sum(int(np.sum(a!=b)) for a,b in zip(n2,e2))
    bad+=b
print('jump mismatches',bad)
