from .clustering_module import ClusteringModule  # noqa: F401
