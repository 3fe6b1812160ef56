from .synthetic import SyntheticReader, synthetic_batch  # noqa: F401
from .indexers import RobertaTokenIndexer, RobertaVocabulary, TokenIndexer, Vocabulary  # noqa: F401
from .readers import DatasetReader, FlattenedGloveGoodNewsReader, NYTimesFacesNERMatchedReader  # noqa: F401
from .iterators import BucketIterator, DataIterator, collate  # noqa: F401
from .shards import read_shard, write_shard  # noqa: F401
