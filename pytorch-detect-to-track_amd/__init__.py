# Directory root of the MI355X-native D&T hot path: csrc/ (HIP kernels + C ABI), lib/ (built
# libdtt_hip.so), dtt/ (Python host side).  Add this directory to sys.path and `import dtt`.
