for tol in 0 1e-12 1e-11 1e-9; do
echo "== PYCOLLO_AMD_KKT_RESID_TOL=$tol"
PYCOLLO_AMD_KKT_RESID_TOL=$tol python tools/ipm_resident_time.py hypersensitive 2000 6 resident 2>&1 | grep -v amdgpu.ids | tail -1
PYCOLLO_AMD_KKT_RESID_TOL=$tol python tools/ipm_resident_time.py cart_pole 5000 4 resident 2>&1 | grep -v amdgpu.ids | tail -1
done
