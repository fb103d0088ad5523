import numpy as np
def lane_sweep(pk, M_, N, NX, NU, off):
    """literal emulation of section B of solve_col_kernel: m[r][lane], 64 lanes"""
    NZ=NX+NU; LD=NZ+1
    PK_G,PK_HD,PK_HXY,PK_HVT,PK_E,PK_C,PK_CF,PK_ZERO=off
    lanes=np.arange(64)
    is_state=lanes<NX; is_ctrl=(lanes>=32)&(lanes<32+NU); lvalid=is_state|is_ctrl
    ms=np.where(is_state,lanes,0); ma=np.where(is_ctrl,lanes-32,0)
    mycol=np.where(is_state,NU+ms,ma); mrob=np.where(is_state,ms//3,ma>>1)
    th_lane=is_state&((ms-3*mrob)==2)
    srcA=np.where(th_lane,3*mrob,np.where(is_ctrl,np.where(ma&1,3*mrob+2,3*mrob),lanes))
    srcB=np.where(th_lane,3*mrob+1,np.where(is_ctrl,np.where(ma&1,3*mrob+2,3*mrob+1),lanes))
    def LC(c): return 32+c if c<NU else c-NU
    hoff=np.zeros((NZ,64),dtype=int)
    for a in range(NZ):
        au=a<NU; sa=0 if au else a-NU; ia=(a>>1) if au else sa//3; da=sa-3*(sa//3)
        h=np.full(64,PK_ZERO)
        if (not au) and da<2:
            dc=ms-3*mrob
            lo=np.minimum(ia,mrob); hi=np.maximum(ia,mrob)
            pe=PK_E+3*(lo*(2*M_-lo-1)//2+(hi-lo-1))+da+dc
            h=np.where(is_state&(dc<2),np.where(ia==mrob,PK_HXY+ia,pe),h)
        if au and (a&1)==0: h=np.where(is_state&(ms==3*ia+2),PK_HVT+ia,h)
        if (not au) and da==2: h=np.where(is_ctrl&(ma==2*ia),PK_HVT+ia,h)
        h=np.where(lvalid&(mycol==a),PK_HD+a,h)
        hoff[a]=np.where(lvalid,h,PK_ZERO)
    goff=np.where(lvalid,PK_G+mycol,PK_ZERO); cfo=PK_CF+3*mycol
    m=np.zeros((NZ+1,64))
    pkN=pk[N]
    for r in range(NX): m[NU+r]=np.where(is_state&(ms==r),pkN[PK_HD+NU+ms],0.0)
    m[NZ]=np.where(is_state,pkN[PK_G+NU+ms],0.0)
    rows={}
    for k in range(N-1,-1,-1):
        PK=pk[k]
        c0,c1,c2=PK[cfo],PK[cfo+1],PK[cfo+2]
        kO=np.where(is_state,c0,0.0); kA=np.where(is_state,c1,np.where(is_ctrl,c0,0.0)); kB=np.where(is_state,c2,np.where(is_ctrl,c1,0.0))
        for s in range(NX): m[NZ]=m[NZ]-m[NU+s]*PK[PK_C+s]
        for r in list(range(NU,NZ))+[NZ]:
            v=m[r].copy(); tA=v[srcA]; tB=v[srcB]
            m[r]=kB*tB+(kA*tA+kO*v)
        for i in range(M_):
            Tc,Ts,Tt=PK[PK_CF+3*(2*i)],PK[PK_CF+3*(2*i)+1],PK[PK_CF+3*(2*i+1)]
            ai,bi=PK[PK_CF+3*(NU+3*i+2)+1],PK[PK_CF+3*(NU+3*i+2)+2]
            gx,gy,gt=m[NU+3*i].copy(),m[NU+3*i+1].copy(),m[NU+3*i+2].copy()
            m[2*i]=Ts*gy+Tc*gx; m[2*i+1]=Tt*gt; m[NU+3*i+2]=bi*gy+(ai*gx+gt)
        for r in range(NZ): m[r]=m[r]+PK[hoff[r]]
        m[NZ]=m[NZ]+PK[goff]
        R_=np.zeros((NU,LD))
        for j in range(NU):
            d=m[j][LC(j)]; inv=1.0/d
            for c in range(NZ): R_[j,c]=m[j][LC(c)]
            rhs_j=m[NZ][LC(j)]; R_[j,NZ]=rhs_j
            rjv=m[j]*inv
            for a in range(j+1,NZ):
                m[a]=m[a]-m[j][LC(a)]*rjv
            m[NZ]=m[NZ]-rhs_j*rjv
        rows[k]=R_
    return rows
