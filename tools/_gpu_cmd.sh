for i in 1 2 3; do
echo "== current"; python tools/exp_r3_modes.py --what step --series factorised 2>&1 | grep step_many
echo "== older (72cbde6)"; MGX_LIB=/root/repo/_ab/pymgrid_amd/libmgx.so python tools/exp_r3_modes.py --what step --series factorised 2>&1 | grep step_many
done
