"""MI355X-native counterpart of external/pointnet2_ops_lib/pointnet2_ops."""
