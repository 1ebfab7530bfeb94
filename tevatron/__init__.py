# Drop-in namespace: only tevatron.faiss_retriever lives here. extend_path keeps the rest of
# an installed Tevatron (driver, modeling, ...) importable when both are on sys.path.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
