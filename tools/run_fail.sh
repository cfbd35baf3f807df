for k in "max_length" "hipgraph" "text_inputs" "beam5_batch256" "ragged"; do
echo "=========== $k"
python -m pytest tests/test_hip_e2e.py -m gpu -q -x -k "$k" 2>&1 | grep -E "Error|error|assert|^E " | head -20 | cut -c1-400
done
