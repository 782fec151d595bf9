"""MI355X-native drop-in for the reference's `simple_knn` package (`from simple_knn._C import distCUDA2`)."""
