"""MI355X-native MultiMAE pretraining step: HIP kernels behind the reference's module API."""
